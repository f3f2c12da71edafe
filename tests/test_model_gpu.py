"""DenseNet-121 embedder on the GPU vs the CPU oracle restatement (fp32).  Tolerance: 1e-5
absolute on unit-norm embeddings, 1e-4 relative on the raw stem/head maps."""
import numpy as np
import pytest
import torch

from oracle import densenet as OD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model_and_sd():
    from mirx.model import DenseNet121
    torch.manual_seed(0)
    m = DenseNet121().eval()
    sd = OD.randomize_bn_stats(m.state_dict(), seed=1)
    m.load_state_dict(sd)
    return m.cuda(), {k: v.cpu() for k, v in sd.items()}


def test_stem_kernel(model_and_sd):
    import ctypes
    from mirx import _lib
    m, sd = model_and_sd
    lib = _lib.load()
    for (b, h, w) in ((3, 224, 224), (2, 64, 96), (1, 36, 20)):
        x = torch.randn(b, 3, h, w, generator=torch.Generator().manual_seed(h))
        ref = OD.stem(x, sd)
        f = m.densenet121[0]
        from mirx.model import _bn_affine
        sc, sh = _bn_affine(f.norm0)
        xg = x.cuda()
        y = torch.empty((b, 64, h // 4, w // 4), device="cuda")
        wt = f.conv0.weight.detach().contiguous()
        rc = lib.mirx_stem_conv7_bn_relu_pool(ctypes.c_void_p(xg.data_ptr()), ctypes.c_void_p(wt.data_ptr()),
                                              ctypes.c_void_p(sc.data_ptr()), ctypes.c_void_p(sh.data_ptr()),
                                              b, h, w, ctypes.c_void_p(y.data_ptr()), None)
        assert rc == 0
        torch.cuda.synchronize()
        torch.testing.assert_close(y.cpu(), ref, atol=2e-4, rtol=1e-4)
        # the three-term bf16 MFMA variant: same map, within 2e-6 of the fp32-MFMA kernel (relative to its maximum)
        from mirx.model import _stem_weights_split3
        w3 = _stem_weights_split3(f.conv0.weight)
        assert w3.shape == (2, 11, 3, 32, 16) and w3.dtype == torch.bfloat16
        y3 = torch.full_like(y, float("nan"))
        rc = lib.mirx_stem_conv7_bn_relu_pool_split3(ctypes.c_void_p(xg.data_ptr()), ctypes.c_void_p(w3.data_ptr()),
                                                     ctypes.c_void_p(sc.data_ptr()), ctypes.c_void_p(sh.data_ptr()),
                                                     b, h, w, ctypes.c_void_p(y3.data_ptr()), None)
        assert rc == 0
        torch.cuda.synchronize()
        torch.testing.assert_close(y3.cpu(), ref, atol=2e-4, rtol=1e-4)
        assert float((y3 - y).abs().max()) < 2e-6 * float(y.abs().max())


def test_head_kernel():
    import ctypes
    from mirx import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    for (b, c, hw) in ((5, 1024, 49), (2, 300, 144), (1, 64, 1)):
        x = torch.randn(b, c, hw, generator=g)
        sc = 0.5 + torch.rand(c, generator=g)
        sh = 0.2 * torch.randn(c, generator=g)
        ref = torch.relu(x * sc[None, :, None] + sh[None, :, None]).mean(dim=2)
        for normalize in (0, 1):
            want = torch.nn.functional.normalize(ref, dim=1) if normalize else ref
            xg, scg, shg = x.cuda(), sc.cuda(), sh.cuda()
            y = torch.empty((b, c), device="cuda")
            rc = lib.mirx_bn_relu_gap_l2norm(ctypes.c_void_p(xg.data_ptr()), ctypes.c_void_p(scg.data_ptr()),
                                             ctypes.c_void_p(shg.data_ptr()), b, c, hw, normalize,
                                             ctypes.c_void_p(y.data_ptr()), None)
            assert rc == 0
            torch.cuda.synchronize()
            torch.testing.assert_close(y.cpu(), want, atol=2e-6, rtol=1e-5)


def test_embeddings_match_cpu_restatement(model_and_sd):
    m, sd = model_and_sd
    x = torch.randn(6, 3, 224, 224, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        y = m(x.cuda()).cpu()
        ref = OD.embed(x, sd)
    assert y.shape == (6, 1024)
    assert float((y.norm(dim=1) - 1).abs().max()) < 1e-6
    assert float((y - ref).abs().max()) <= 1e-5, float((y - ref).abs().max())
    # eager torch path on the GPU (grad enabled) agrees too
    y2 = m(x[:2].cuda()).detach().cpu()
    assert float((y2 - ref[:2]).abs().max()) <= 1e-5
    # fc / classification-head variants keep the reference's output contract
    from mirx.model import DenseNet121
    torch.manual_seed(1)
    m2 = DenseNet121(embedding_dim=256, num_labels=3).eval().cuda()
    with torch.no_grad():
        out = m2(x[:2].cuda())
    assert out["embedding"].shape == (2, 256) and out["logits"].shape == (2, 3)


def test_fused_elementwise_kernels():
    import ctypes
    from mirx import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(9)
    for (b, ctot, c, h, w) in ((3, 256, 96, 56, 56), (2, 1024, 992, 7, 7), (2, 64, 64, 14, 14)):
        buf = torch.randn(b, ctot, h, w, generator=g)
        sc = 0.5 + torch.rand(c, generator=g)
        sh = 0.3 * torch.randn(c, generator=g)
        want = torch.relu(buf[:, :c] * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
        bg, scg, shg = buf.cuda(), sc.cuda(), sh.cuda()
        y = torch.empty((b, c, h, w), device="cuda")
        assert lib.mirx_bn_relu_nchw(ctypes.c_void_p(bg.data_ptr()), ctot * h * w, ctypes.c_void_p(scg.data_ptr()),
                                     ctypes.c_void_p(shg.data_ptr()), b, c, h * w,
                                     ctypes.c_void_p(y.data_ptr()), None) == 0
        torch.cuda.synchronize()
        torch.testing.assert_close(y.cpu(), want, atol=1e-6, rtol=1e-6)
        if h % 2 == 0:
            want_p = torch.nn.functional.avg_pool2d(want, 2, 2)
            yp = torch.empty((b, c, h // 2, w // 2), device="cuda")
            assert lib.mirx_bn_relu_avgpool2(ctypes.c_void_p(bg.data_ptr()), ctot * h * w,
                                             ctypes.c_void_p(scg.data_ptr()), ctypes.c_void_p(shg.data_ptr()),
                                             b, c, h, w, ctypes.c_void_p(yp.data_ptr()), 0, None) == 0
            torch.cuda.synchronize()
            torch.testing.assert_close(yp.cpu(), want_p, atol=1e-6, rtol=1e-6)


def test_fused_conv1x1_kernel():
    """mirx_conv1x1_bn_relu against torch: prologue/epilogue variants, 7x7 maps (scalar staging path),
    partial last tiles, channel-prefix views of a wider buffer."""
    import ctypes
    from mirx import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    for (b, ctot, cin, hw_shape, cout, prologue, relu_out) in (
            (3, 256, 96, (56, 56), 128, True, True), (5, 1024, 992, (7, 7), 128, True, True),
            (2, 512, 512, (14, 14), 256, False, False), (2, 64, 64, (10, 6), 128, True, False)):
        h, w = hw_shape
        buf = torch.randn(b, ctot, h, w, generator=g)
        wt4 = 0.1 * torch.randn(cout, cin, 1, 1, generator=g)
        sc, sh = 0.5 + torch.rand(cin, generator=g), 0.3 * torch.randn(cin, generator=g)
        bias = 0.2 * torch.randn(cout, generator=g)
        xin = buf[:, :cin]
        if prologue:
            xin = torch.relu(xin * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
        want = torch.nn.functional.conv2d(xin, wt4, bias)
        if relu_out:
            want = torch.relu(want)
        bg = buf.cuda()
        wtg = wt4.view(cout, cin).t().contiguous().cuda()
        scg, shg, biasg = sc.cuda(), sh.cuda(), bias.cuda()
        y = torch.empty((b, cout, h, w), device="cuda")
        rc = lib.mirx_conv1x1_bn_relu(ctypes.c_void_p(bg.data_ptr()), ctot * h * w, cin,
                                      ctypes.c_void_p(scg.data_ptr()) if prologue else None,
                                      ctypes.c_void_p(shg.data_ptr()) if prologue else None,
                                      ctypes.c_void_p(wtg.data_ptr()), ctypes.c_void_p(biasg.data_ptr()), b, h * w, cout,
                                      1 if relu_out else 0, ctypes.c_void_p(y.data_ptr()), None)
        assert rc == 0, lib.mirx_last_error()
        torch.cuda.synchronize()
        torch.testing.assert_close(y.cpu(), want, atol=1e-4, rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("side,batch", [(56, 3), (28, 5), (14, 9), (7, 6), (7, 5)])
def test_winograd_conv3x3_matches_direct_conv(side, batch):
    """mirx_conv3x3_winograd_nchw (conv2 of a dense layer, model.py:53) against a float64 direct
    convolution, written into a channel slice of a wider buffer.  Winograd F(2x2,3x3) in fp32: tolerance
    2e-5 relative to the largest output (the same transform MIOpen's library kernel uses)."""
    import ctypes
    from mirx import _lib
    from mirx.model import _winograd_weights
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(side)
    x = torch.randn(batch, 128, side, side, generator=g, device=dev)
    w = torch.randn(32, 128, 3, 3, generator=g, device=dev) * 0.05
    buf = torch.full((batch, 96, side, side), 7.0, device=dev)
    c0 = 40
    u = _winograd_weights(w)
    assert u.shape == (16, 16, 8, 32)
    _lib.check(lib.mirx_conv3x3_winograd_nchw(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(u.data_ptr()), batch, side,
                                              ctypes.c_void_p(buf.data_ptr() + 4 * c0 * side * side), 96 * side * side,
                                              None), "conv3x3")
    torch.cuda.synchronize()
    want = torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), None, padding=1)
    got = buf[:, c0:c0 + 32].double().cpu()
    assert float((got - want).abs().max()) < 2e-5 * float(want.abs().max())
    assert bool((buf[:, :c0] == 7.0).all()) and bool((buf[:, c0 + 32:] == 7.0).all())      # neighbours untouched
    rc = lib.mirx_conv3x3_winograd_nchw(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(u.data_ptr()), batch, 12,
                                        ctypes.c_void_p(buf.data_ptr()), 96 * side * side, None)
    assert rc == -1


@pytest.mark.gpu
@pytest.mark.parametrize("side,batch", [(56, 3), (28, 5), (14, 9), (14, 1)])
def test_winograd_split3_conv3x3_matches_direct_conv(side, batch):
    """mirx_conv3x3_winograd_split3_nchw (Winograd-domain GEMMs on three-term bf16 MFMAs) against a float64
    direct convolution and against the fp32-MFMA kernel: same tolerance as the fp32 kernel (2e-5 of the largest
    output, the Winograd transform's own error), and within 3e-6 of that kernel (the split's own error)."""
    import ctypes
    from mirx import _lib
    from mirx.model import _winograd_weights, _winograd_weights_split3
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(side + 100)
    x = torch.randn(batch, 128, side, side, generator=g, device=dev)
    x = torch.relu(x) * 1.7                                        # post-ReLU statistics, like conv2's real input
    w = torch.randn(32, 128, 3, 3, generator=g, device=dev) * 0.05
    buf = torch.full((batch, 96, side, side), 7.0, device=dev)
    ref = torch.full((batch, 96, side, side), 7.0, device=dev)
    c0 = 40
    u3 = _winograd_weights_split3(w)
    assert u3.shape == (8, 16, 3, 32, 16) and u3.dtype == torch.bfloat16
    vp = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + off)      # noqa: E731
    _lib.check(lib.mirx_conv3x3_winograd_split3_nchw(vp(x), vp(u3), batch, side, vp(buf, 4 * c0 * side * side),
                                                     96 * side * side, None), "conv3x3_split3")
    _lib.check(lib.mirx_conv3x3_winograd_nchw(vp(x), vp(_winograd_weights(w)), batch, side, vp(ref, 4 * c0 * side * side),
                                              96 * side * side, None), "conv3x3")
    torch.cuda.synchronize()
    want = torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), None, padding=1)
    got = buf[:, c0:c0 + 32].double().cpu()
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) < 2e-5 * scale
    assert float((buf[:, c0:c0 + 32] - ref[:, c0:c0 + 32]).abs().max()) < 3e-6 * scale
    assert bool((buf[:, :c0] == 7.0).all()) and bool((buf[:, c0 + 32:] == 7.0).all())      # neighbours untouched
    assert lib.mirx_conv3x3_winograd_split3_nchw(vp(x), vp(u3), batch, 7, vp(buf), 96 * side * side, None) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("side,batch", [(56, 3), (28, 5), (14, 9), (14, 1)])
def test_direct_split3_conv3x3_matches_direct_conv(side, batch):
    """mirx_conv3x3_direct_split3_nchw (implicit GEMM over 9 taps x 128 channels on three-term bf16 MFMAs) against a
    float64 direct convolution: 3e-6 of the largest output (no Winograd transform error: tighter than the Winograd
    kernels' 2e-5), written into a channel slice of a wider buffer."""
    import ctypes
    from mirx import _lib
    from mirx.model import _conv3x3_weights_split3
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(side + 200)
    x = torch.relu(torch.randn(batch, 128, side, side, generator=g, device=dev)) * 1.7
    w = torch.randn(32, 128, 3, 3, generator=g, device=dev) * 0.05
    buf = torch.full((batch, 96, side, side), 7.0, device=dev)
    c0 = 40
    w3 = _conv3x3_weights_split3(w)
    assert w3.shape == (8, 9, 3, 32, 16) and w3.dtype == torch.bfloat16
    vp = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + off)      # noqa: E731
    _lib.check(lib.mirx_conv3x3_direct_split3_nchw(vp(x), vp(w3), batch, side, vp(buf, 4 * c0 * side * side),
                                                   96 * side * side, None), "conv3x3_direct")
    torch.cuda.synchronize()
    want = torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), None, padding=1)
    got = buf[:, c0:c0 + 32].double().cpu()
    assert float((got - want).abs().max()) < 3e-6 * float(want.abs().max())
    assert bool((buf[:, :c0] == 7.0).all()) and bool((buf[:, c0 + 32:] == 7.0).all())      # neighbours untouched
    assert lib.mirx_conv3x3_direct_split3_nchw(vp(x), vp(w3), batch, 7, vp(buf), 96 * side * side, None) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(256, 256), (160, 192), (112, 112)])
def test_embeddings_at_other_resolutions(model_and_sd, size):
    """Inputs whose feature maps are not 56/28/14/7 (the reference resizes to 224, read_data.py, but the
    module accepts any size): the specialised kernels must step aside or generalise, never mis-index.
    112x112 exercises the Winograd kernel on the 28- and 14-wide maps of blocks 1 and 2."""
    m, sd = model_and_sd
    x = torch.randn(3, 3, size[0], size[1], generator=torch.Generator().manual_seed(size[0]))
    with torch.no_grad():
        y = m(x.cuda()).cpu()
        ref = OD.embed(x, sd)
    assert float((y - ref).abs().max()) <= 1e-5, float((y - ref).abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,hw,n,prologue,relu", [(256, 128, 196, 3, True, True), (512, 128, 49, 5, True, True),
                                                         (1024, 512, 49, 2, False, False), (64, 128, 3136, 1, True, True),
                                                         (992, 128, 37, 3, True, True)])
def test_split3_conv1x1_matches_float64(cin, cout, hw, n, prologue, relu):
    """mirx_conv1x1_bn_relu_split3 (three bf16 terms per operand, six MFMAs per product) against a float64
    restatement of relu(W relu(bn(x)) + b): fp32-grade -- 3e-6 relative to the largest output, the same
    bound the fp32-MFMA kernel meets."""
    import ctypes
    from mirx import _lib
    from mirx.model import _split3_weights
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(cin + hw)
    ctot = cin + 32
    buf = torch.randn(n, ctot, hw, generator=g, device=dev)
    w = torch.randn(cout, cin, generator=g, device=dev) / cin ** 0.5
    sc = torch.rand(cin, generator=g, device=dev) + 0.5
    sh = torch.randn(cin, generator=g, device=dev) * 0.3
    bias = torch.randn(cout, generator=g, device=dev)
    w3 = _split3_weights(w)
    assert w3.shape == (cout // 128, cin // 16, 3, 128, 16) and w3.dtype == torch.bfloat16
    ybuf = torch.full((n, cout + 8, hw), -5.0, device=dev)          # written as a channel prefix of a wider buffer
    y = ybuf[:, :cout]
    vp = lambda t: ctypes.c_void_p(t.data_ptr())                       # noqa: E731
    _lib.check(lib.mirx_conv1x1_bn_relu_split3(vp(buf), ctot * hw, cin, vp(sc) if prologue else None,
                                               vp(sh) if prologue else None, vp(w3), vp(bias), n, hw, cout,
                                               1 if relu else 0, vp(y), (cout + 8) * hw, None), "split3")
    torch.cuda.synchronize()
    xin = buf[:, :cin].double()
    if prologue:
        xin = torch.relu(xin * sc.double()[None, :, None] + sh.double()[None, :, None])
    want = torch.einsum("oc,bcp->bop", w.double(), xin) + bias.double()[None, :, None]
    if relu:
        want = torch.relu(want)
    err = float((y.double() - want).abs().max())
    assert err < 3e-6 * max(1.0, float(want.abs().max())), err
    assert bool((ybuf[:, cout:] == -5.0).all())


@pytest.fixture(params=["tiled", "small"])
def conv1x1_kernel(request):
    """Both kernels behind mirx_conv1x1_bn_relu_split2h[_terms]: k_conv1x1_h2 (128 x 128 tiles through LDS) and k_conv1x1_h2s
    (one wave per 32 x 32 tile, the small-launch kernel) -- selected by launch size in production, forced here."""
    from mirx import _lib
    lib = _lib.load()
    for key in (_lib.TUNE_CONV1X1_SMALL_MAX_WG, _lib.TUNE_CONV3X3_SMALL_MAX_WG):       # the 3x3 conv has the same pair of kernels
        _lib.check(lib.mirx_set_tuning(key, 0 if request.param == "tiled" else 1 << 20), "set_tuning")
    yield request.param
    _lib.check(lib.mirx_set_tuning(_lib.TUNE_CONV1X1_SMALL_MAX_WG, 128), "set_tuning")
    _lib.check(lib.mirx_set_tuning(_lib.TUNE_CONV3X3_SMALL_MAX_WG, 96), "set_tuning")



@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,hw,n,prologue,relu,mag", [(256, 128, 196, 3, True, True, 1.0), (512, 128, 49, 5, True, True, 40.0),
                                                             (1024, 512, 49, 2, False, False, 1e-3), (64, 128, 3136, 1, True, True, 300.0),
                                                             (992, 128, 37, 3, True, True, 1.0)])
def test_split2h_conv1x1_matches_float64(cin, cout, hw, n, prologue, relu, mag, conv1x1_kernel):
    """mirx_conv1x1_bn_relu_split2h (two fp16 terms per operand, three MFMAs per product; the range of every image read from
    the input's range row) against a float64 restatement: fp32-grade (3e-6 relative to the largest output) at any input
    magnitude, every image's output range published exactly, bytes beyond the channel prefix untouched; a non-finite range
    poisons THAT image only and leaves the others bit-identical."""
    import ctypes
    from mirx import _lib
    from mirx.model import _split2h_weights
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(cin + hw)
    ctot = cin + 32
    buf = torch.randn(n, ctot, hw, generator=g, device=dev) * mag
    w = torch.randn(cout, cin, generator=g, device=dev) / cin ** 0.5
    w[3] *= 1e-4                                                    # one output channel with tiny weights: its own scale
    sc = torch.rand(cin, generator=g, device=dev) + 0.5
    sh = torch.randn(cin, generator=g, device=dev) * 0.3 * mag
    bias = torch.randn(cout, generator=g, device=dev) * mag
    w2, osc = _split2h_weights(w)
    assert w2.shape == (cout // 128, cin // 16, 2, 128, 16) and w2.dtype == torch.float16 and osc.shape == (cout,)
    slots_in = buf[:, :cin].abs().amax(dim=(1, 2)).contiguous()    # what the producers of the prefix published, per image
    slots_out = torch.zeros(n, device=dev)
    ybuf = torch.full((n, cout + 8, hw), -5.0, device=dev)
    y = ybuf[:, :cout]
    vp = lambda t: ctypes.c_void_p(t.data_ptr())                       # noqa: E731
    ks, kb = (float(sc.abs().max()), float(sh.abs().max())) if prologue else (1.0, 0.0)
    _lib.check(lib.mirx_conv1x1_bn_relu_split2h(vp(buf), ctot * hw, cin, vp(sc) if prologue else None,
                                                vp(sh) if prologue else None, vp(w2), vp(osc), vp(bias), n, hw, cout,
                                                1 if relu else 0, vp(y), (cout + 8) * hw, vp(slots_in), ks, kb,
                                                vp(slots_out), 0, 0, None), "split2h")
    torch.cuda.synchronize()
    xin = buf[:, :cin].double()
    if prologue:
        xin = torch.relu(xin * sc.double()[None, :, None] + sh.double()[None, :, None])
    want = torch.einsum("oc,bcp->bop", w.double(), xin) + bias.double()[None, :, None]
    if relu:
        want = torch.relu(want)
    err = float((y.double() - want).abs().max())
    assert err < 3e-6 * max(mag, float(want.abs().max())), err
    assert bool((ybuf[:, cout:] == -5.0).all())
    assert torch.equal(slots_out, y.abs().amax(dim=(1, 2)))
    # a non-finite range poisons that image's output instead of returning finite garbage -- and only that image's
    good = y.clone()
    slots_in[n - 1] = float("inf")
    _lib.check(lib.mirx_conv1x1_bn_relu_split2h(vp(buf), ctot * hw, cin, vp(sc) if prologue else None,
                                                vp(sh) if prologue else None, vp(w2), vp(osc), vp(bias), n, hw, cout,
                                                1 if relu else 0, vp(y), (cout + 8) * hw, vp(slots_in), ks, kb, None, 0, 0, None),
               "split2h")
    torch.cuda.synchronize()
    assert bool(torch.isnan(y[n - 1]).all())
    assert torch.equal(y[:n - 1], good[:n - 1])


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,hw,n,terms", [(64, 128, 3136, 1, True), (256, 128, 784, 3, True), (512, 128, 196, 5, True),
                                                 (1008, 128, 196, 1, True), (992, 128, 49, 7, True), (512, 128, 49, 1, True),
                                                 (256, 128, 784, 2, False), (512, 256, 196, 3, False), (1024, 512, 49, 5, False),
                                                 (128, 128, 37, 9, True), (320, 128, 196, 300, True)])
def test_small_launch_conv1x1_is_bit_identical_to_the_tiled_kernel(cin, cout, hw, n, terms):
    """k_conv1x1_h2s (one wave per 32 x 32 tile, no LDS: the reference's batch sizes, test.py:1513, milvus_retrieval.py:53-66)
    against k_conv1x1_h2 on the same input: the same bits in every output value, published range and bottleneck scale -- so a
    row's embedding does not depend on which kernel its batch size selected.  Images of different magnitudes; a poisoned one is
    NaN from both kernels."""
    import ctypes
    from mirx import _lib
    from mirx.model import _split2h_weights
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(cin * 7 + hw + n)
    ctot = cin + 32
    buf = torch.randn(n, ctot, hw, generator=g, device=dev)
    buf *= (10.0 ** torch.randint(-3, 3, (n, 1, 1), generator=g, device=dev).float())
    w = torch.randn(cout, cin, generator=g, device=dev) / cin ** 0.5
    sc = torch.rand(cin, generator=g, device=dev) + 0.5
    sh = torch.randn(cin, generator=g, device=dev) * 0.3
    bias = torch.randn(cout, generator=g, device=dev)
    w2, osc = _split2h_weights(w)
    rng_in = buf[:, :cin].abs().amax(dim=(1, 2)).contiguous()
    if n > 2:
        rng_in[1] = float("inf")                                       # a poisoned image: NaN bits must agree too
    vp = lambda t: ctypes.c_void_p(t.data_ptr())                       # noqa: E731
    outs = []
    try:
        for limit in (0, 1 << 20):
            _lib.check(lib.mirx_set_tuning(_lib.TUNE_CONV1X1_SMALL_MAX_WG, limit), "set_tuning")
            y = torch.full((n, cout, hw), -7.0, device=dev)
            aux = torch.zeros(n, device=dev)
            if terms:
                _lib.check(lib.mirx_conv1x1_bn_relu_split2h_terms(vp(buf), ctot * hw, cin, vp(sc), vp(sh), vp(w2), vp(osc), vp(bias), n, hw,
                                                                  vp(y), vp(rng_in), float(sc.abs().max()), float(sh.abs().max()),
                                                                  float(w.abs().sum(dim=1).max()), float(bias.abs().max()), vp(aux),
                                                                  0, None), "terms")
            else:
                _lib.check(lib.mirx_conv1x1_bn_relu_split2h(vp(buf), ctot * hw, cin, vp(sc), vp(sh), vp(w2), vp(osc), vp(bias), n, hw, cout,
                                                            1, vp(y), cout * hw, vp(rng_in), float(sc.abs().max()),
                                                            float(sh.abs().max()), vp(aux), 0, 0, None), "split2h")
            torch.cuda.synchronize()
            outs.append((y.view(torch.int32).clone(), aux.view(torch.int32).clone()))
    finally:
        _lib.check(lib.mirx_set_tuning(_lib.TUNE_CONV1X1_SMALL_MAX_WG, 128), "set_tuning")
    clean = [b for b in range(n) if not (n > 2 and b == 1)]
    assert torch.equal(outs[0][0][clean], outs[1][0][clean])
    assert torch.equal(outs[0][1][clean], outs[1][1][clean])
    if n > 2:                                                          # the poisoned image: NaN from both (payload bits are not a contract)
        for y, aux in outs:
            # terms: the image's 2^-t is NaN, which is what poisons the 3x3 conv that consumes the (then meaningless) terms
            assert bool(torch.isnan(aux.view(torch.float32)[1])) if terms else bool(torch.isnan(y[1].view(torch.float32)).all())


@pytest.mark.gpu
@pytest.mark.parametrize("side,n", [(56, 1), (28, 3), (14, 5), (14, 1), (7, 9), (7, 1), (14, 130)])
def test_small_launch_conv3x3_is_bit_identical_to_the_strip_kernel(side, n):
    """k_conv3x3_d2s (one wave per 32 output pixels, no LDS) against k_conv3x3_d2p on the same pre-split bottleneck: the same
    bits in every output value and published range."""
    import ctypes
    from mirx import _lib
    from mirx.model import YTERMS_CHANNEL_ORDER, _conv3x3_weights_split2h, _split2h_weights
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(side * 31 + n)
    cin, hw = 64, side * side
    buf = torch.randn(n, cin, hw, generator=g, device=dev)
    buf *= (10.0 ** torch.randint(-2, 3, (n, 1, 1), generator=g, device=dev).float())
    w1 = torch.randn(128, cin, generator=g, device=dev) / cin ** 0.5
    sc = torch.rand(cin, generator=g, device=dev) + 0.5
    sh = torch.randn(cin, generator=g, device=dev) * 0.3
    b1 = torch.randn(128, generator=g, device=dev) * 0.2
    w3 = torch.randn(32, 128, 3, 3, generator=g, device=dev) * 0.05
    w2, osc = _split2h_weights(w1)
    c3, c3osc = _conv3x3_weights_split2h(w3, YTERMS_CHANNEL_ORDER)
    y = torch.empty((n, 128, hw), device=dev)
    rng = buf.abs().amax(dim=(1, 2)).contiguous()
    yinv = torch.zeros(n, device=dev)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())                       # noqa: E731
    _lib.check(lib.mirx_conv1x1_bn_relu_split2h_terms(vp(buf), cin * hw, cin, vp(sc), vp(sh), vp(w2), vp(osc), vp(b1), n, hw, vp(y),
                                                      vp(rng), float(sc.abs().max()), float(sh.abs().max()),
                                                      float(w1.abs().sum(dim=1).max()), float(b1.abs().max()), vp(yinv), 0, None), "terms")
    outs = []
    try:
        for limit in (0, 1 << 20):
            _lib.check(lib.mirx_set_tuning(_lib.TUNE_CONV3X3_SMALL_MAX_WG, limit), "set_tuning")
            out = torch.full((n, 40, hw), -3.0, device=dev)
            orng = torch.zeros(n, device=dev)
            _lib.check(lib.mirx_conv3x3_direct_terms_nchw(vp(y), vp(c3), vp(c3osc), n, side, vp(out), 40 * hw, vp(yinv), vp(orng), 0, None),
                       "conv3x3_terms")
            torch.cuda.synchronize()
            assert bool((out[:, 32:] == -3.0).all())
            outs.append((out.view(torch.int32).clone(), orng.view(torch.int32).clone()))
    finally:
        _lib.check(lib.mirx_set_tuning(_lib.TUNE_CONV3X3_SMALL_MAX_WG, 96), "set_tuning")
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    y64 = torch.relu(torch.einsum("oc,bcp->bop", w1.double(), torch.relu(buf.double() * sc.double()[None, :, None] + sh.double()[None, :, None]))
                     + b1.double()[None, :, None]).unflatten(2, (side, side))
    want = torch.nn.functional.conv2d(y64, w3.double(), None, padding=1).flatten(2)
    got = outs[1][0].view(torch.float32)[:, :32].double()
    for b in range(n):
        assert float((got[b] - want[b]).abs().max()) < 3e-6 * float(want[b].abs().max()), b


@pytest.mark.gpu
@pytest.mark.parametrize("side,n,ps_pad", [(56, 3, 0), (28, 5, 16), (14, 7, 0)])
def test_pooled_twin_of_the_3x3_conv_is_bit_identical_to_the_pooling_pass(side, n, ps_pad):
    """mirx_conv3x3_direct_terms_nchw_pool: the transition's norm + relu + avgpool2 of the 32 new channels, written by the 3x3
    launch itself, against mirx_bn_relu_avgpool2 run on the stored outputs -- the same bits; the conv outputs and ranges are the
    plain launch's; a poisoned image stays in its own rows; channels of the pooled map the launch does not own are untouched;
    the entry point refuses launches that take the one-wave-per-block kernel."""
    import ctypes
    from mirx import _lib
    from mirx.model import YTERMS_CHANNEL_ORDER, _conv3x3_weights_split2h, _split2h_weights
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(side * 7 + n)
    cin, hw, hw2, ctot, c0 = 64, side * side, (side // 2) ** 2, 104, 64      # the launch owns channels 64..95 of a 104-channel block
    ps = hw + ps_pad                                                        # padded channel planes of the block's buffer
    buf = torch.randn(n, cin, hw, generator=g, device=dev)
    buf *= (10.0 ** torch.randint(-2, 3, (n, 1, 1), generator=g, device=dev).float())
    buf[1, 3, 5] = float("nan")                                             # image 1 is poisoned
    w1 = torch.randn(128, cin, generator=g, device=dev) / cin ** 0.5
    sc = torch.rand(cin, generator=g, device=dev) + 0.5
    sh = torch.randn(cin, generator=g, device=dev) * 0.3
    b1 = torch.randn(128, generator=g, device=dev) * 0.2
    w3 = torch.randn(32, 128, 3, 3, generator=g, device=dev) * 0.05
    tsc = torch.randn(ctot, generator=g, device=dev)                        # the transition's folded norm (either sign)
    tsh = torch.randn(ctot, generator=g, device=dev) * 0.5
    w2, osc = _split2h_weights(w1)
    c3, c3osc = _conv3x3_weights_split2h(w3, YTERMS_CHANNEL_ORDER)
    y = torch.empty((n, 128, hw), device=dev)
    rng = buf.abs().amax(dim=(1, 2)).contiguous()
    yinv = torch.zeros(n, device=dev)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())                            # noqa: E731
    off = lambda t, k: ctypes.c_void_p(t.data_ptr() + 4 * k)                # noqa: E731
    _lib.check(lib.mirx_conv1x1_bn_relu_split2h_terms(vp(buf), cin * hw, cin, vp(sc), vp(sh), vp(w2), vp(osc), vp(b1), n, hw, vp(y),
                                                      vp(rng), float(sc.abs().max()), float(sh.abs().max()),
                                                      float(w1.abs().sum(dim=1).max()), float(b1.abs().max()), vp(yinv), 0, None), "terms")
    try:
        _lib.check(lib.mirx_set_tuning(_lib.TUNE_CONV3X3_SMALL_MAX_WG, 0), "set_tuning")       # strip kernel whatever the size
        assert lib.mirx_conv3x3_small_launch(n, side) == 0
        plain = torch.full((n, ctot, ps), -3.0, device=dev)
        prng = torch.zeros(n, device=dev)
        _lib.check(lib.mirx_conv3x3_direct_terms_nchw(vp(y), vp(c3), vp(c3osc), n, side, off(plain, c0 * ps), ctot * ps, vp(yinv),
                                                      vp(prng), ps, None), "conv3x3_terms")
        want = torch.full((n, ctot, hw2), -5.0, device=dev)
        _lib.check(lib.mirx_bn_relu_avgpool2_into(off(plain, c0 * ps), ctot * ps, off(tsc, c0), off(tsh, c0), n, 32, side, side,
                                                  off(want, c0 * hw2), ctot * hw2, ps, None), "pool_into")
        out = torch.full((n, ctot, ps), -3.0, device=dev)
        orng = torch.zeros(n, device=dev)
        pooled = torch.full((n, ctot, hw2), -5.0, device=dev)
        _lib.check(lib.mirx_conv3x3_direct_terms_nchw_pool(vp(y), vp(c3), vp(c3osc), n, side, off(out, c0 * ps), ctot * ps, vp(yinv),
                                                           vp(orng), ps, off(tsc, c0), off(tsh, c0), off(pooled, c0 * hw2),
                                                           ctot * hw2, None), "conv3x3_terms_pool")
        torch.cuda.synchronize()
        _lib.check(lib.mirx_set_tuning(_lib.TUNE_CONV3X3_SMALL_MAX_WG, 1 << 20), "set_tuning")  # ... and now the small kernel
        assert lib.mirx_conv3x3_small_launch(n, side) == 1
        assert lib.mirx_conv3x3_direct_terms_nchw_pool(vp(y), vp(c3), vp(c3osc), n, side, off(out, c0 * ps), ctot * ps, vp(yinv),
                                                       vp(orng), ps, off(tsc, c0), off(tsh, c0), off(pooled, c0 * hw2),
                                                       ctot * hw2, None) != 0
        assert b"pooled twin" in lib.mirx_last_error()
    finally:
        _lib.check(lib.mirx_set_tuning(_lib.TUNE_CONV3X3_SMALL_MAX_WG, 96), "set_tuning")
    assert torch.equal(out.view(torch.int32), plain.view(torch.int32)) and torch.equal(orng.view(torch.int32), prng.view(torch.int32))
    assert torch.equal(pooled.view(torch.int32), want.view(torch.int32))
    assert bool((pooled[:, :c0] == -5.0).all()) and bool((pooled[:, c0 + 32:] == -5.0).all())
    ref = torch.nn.functional.avg_pool2d(torch.relu(out[:, c0:c0 + 32, :hw].double() * tsc[c0:c0 + 32].double()[None, :, None]
                                                    + tsh[c0:c0 + 32].double()[None, :, None]).unflatten(2, (side, side)), 2).flatten(2)
    for b in range(n):
        if b == 1:
            continue                                                        # (NaN conv outputs: relu's max drops them, as the pass does)
        assert float((pooled[b, c0:c0 + 32].double() - ref[b]).abs().max()) < 1e-6 * float(ref[b].abs().max() + 1e-30), b


@pytest.mark.gpu
@pytest.mark.parametrize("c,side,n,pad_in,pad_out", [(256, 28, 3, 16, 28), (512, 14, 2, 28, 0), (256, 56, 1, 0, 16)])
def test_transition_kernels_with_padded_planes(c, side, n, pad_in, pad_out, conv1x1_kernel):
    """The two launches of a transition on buffers whose channel planes are padded (mirx.model._plane_stride):
    mirx_bn_relu_avgpool2 reads planes `x_plane_stride` apart, mirx_conv1x1_bn_relu_split2h writes planes `y_plane_stride`
    apart; the gaps hold NaN before and after (never read, never written)."""
    import ctypes
    from mirx import _lib
    from mirx.model import _split2h_weights
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(c + side)
    hw, hw2 = side * side, (side // 2) ** 2
    ps, ps2 = hw + pad_in, hw2 + pad_out
    cout = c // 2
    xs = torch.full((n, c, ps), float("nan"), device=dev)
    x = xs[:, :, :hw].unflatten(2, (side, side))
    x.copy_(torch.randn(n, c, side, side, generator=g, device=dev))
    sc = torch.rand(c, generator=g, device=dev) + 0.5
    sh = torch.randn(c, generator=g, device=dev) * 0.3
    wt = torch.randn(cout, c, generator=g, device=dev) / c ** 0.5
    w2, osc = _split2h_weights(wt)
    pooled = torch.empty((n, c, side // 2, side // 2), device=dev)
    ys = torch.full((n, cout + 32, ps2), float("nan"), device=dev)
    rng_in = x.abs().amax(dim=(1, 2, 3)).contiguous()
    rng_out = torch.zeros(n, device=dev)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())                       # noqa: E731
    _lib.check(lib.mirx_bn_relu_avgpool2(vp(xs), c * ps, vp(sc), vp(sh), n, c, side, side, vp(pooled), ps if pad_in else 0, None),
               "avgpool")
    _lib.check(lib.mirx_conv1x1_bn_relu_split2h(vp(pooled), c * hw2, c, None, None, vp(w2), vp(osc), None, n, hw2, cout, 0, vp(ys),
                                                (cout + 32) * ps2, vp(rng_in), float(sc.abs().max()), float(sh.abs().max()),
                                                vp(rng_out), 0, ps2 if pad_out else 0, None), "conv")
    torch.cuda.synchronize()
    act = torch.relu(x.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    want_p = torch.nn.functional.avg_pool2d(act, 2)
    assert float((pooled.double() - want_p).abs().max()) < 1e-6
    want = torch.einsum("oc,bchw->bohw", wt.double(), want_p).flatten(2)
    got = ys[:, :cout, :hw2].double()
    assert float((got - want).abs().max()) < 3e-6 * max(1.0, float(want.abs().max()))
    assert bool(torch.isnan(ys[:, cout:]).all()) and bool(torch.isnan(ys[:, :, hw2:]).all())
    assert bool(torch.isnan(xs[:, :, hw:]).all())
    assert torch.equal(rng_out, ys[:, :cout, :hw2].abs().amax(dim=(1, 2)))
    # a plane stride below the plane is refused
    assert lib.mirx_bn_relu_avgpool2(vp(xs), c * ps, vp(sc), vp(sh), n, c, side, side, vp(pooled), hw - 4, None) != 0


@pytest.mark.gpu
def test_split2h_path_matches_oracle_and_legacy(model_and_sd):
    """The two-fp16-term DenseNet path (default at 224 x 224) against the CPU restatement (1e-5 on unit-norm
    embeddings) and against the three-bf16-term path of round 1 on the same weights; inputs of very different
    magnitudes (the ranges travel with the data); rows independent of what else is in the batch."""
    import mirx.model as mm
    m, sd = model_and_sd
    x = torch.randn(5, 3, 224, 224, generator=torch.Generator().manual_seed(77))
    x[1] *= 30.0
    x[2] *= 1e-3
    with torch.no_grad():
        assert m._h2_ok(x.cuda())
        e2 = m(x.cuda()).cpu()
        old = m.configure(densenet_two_fp16=False)
        try:
            e3 = m(x.cuda()).cpu()
        finally:
            m.configure(**old.__dict__)
        ref = OD.embed(x, sd)
    per = lambda a, b: [f"{v:.1e}" for v in (a - b).abs().amax(1).tolist()]      # noqa: E731
    assert float((e2 - ref).abs().max()) <= 1e-5, (per(e2, ref), per(e3, ref), per(e2, e3))
    assert float((e2 - e3).abs().max()) <= 2e-6, (per(e2, ref), per(e3, ref), per(e2, e3))
    with torch.no_grad():
        solo = m(x[2:3].cuda()).cpu()                   # ranges are per image: alone or in a batch, the same bits
    assert torch.equal(solo, e2[2:3])


@pytest.mark.gpu
def test_concurrent_forwards_on_two_streams_equal_the_sequential_result(model_and_sd):
    """bench.py embeds the two halves of a micro-batch concurrently on two HIP streams (same module, shared weight caches).
    The halves must come out bit-identical to the same halves embedded one after the other on the default stream -- a
    missing dependency between the streams (range slots, scratch buffers, cached weights) would show here."""
    m, _ = model_and_sd
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(512, 3, 224, 224, generator=g, device="cuda")
    x[300:] *= 7.0                                             # different ranges in the two halves
    with torch.no_grad():
        seq = torch.cat([m(x[:256]), m(x[256:])], 0)
        torch.cuda.synchronize()
        out = torch.empty_like(seq)
        cur = torch.cuda.current_stream()
        side = [torch.cuda.Stream(), torch.cuda.Stream()]
        for rep in range(3):                                   # a few rounds: the streams drift against each other
            for j, st in enumerate(side):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    out[256 * j:256 * (j + 1)] = m(x[256 * j:256 * (j + 1)])
            for st in side:
                cur.wait_stream(st)
            torch.cuda.synchronize()
            assert torch.equal(out, seq), rep


@pytest.mark.gpu
def test_padded_channel_planes_agree(model_and_sd):
    """Padded channel planes (KernelConfig.plane_stride: measured, no gain, kept as an option of the kernels' ABI) against the
    packed default on the same weights: the same bits, gaps never read."""
    import mirx.model as mm
    m, sd = model_and_sd
    x = torch.randn(5, 3, 224, 224, generator=torch.Generator().manual_seed(5))
    x[3] *= 12.0
    with torch.no_grad():
        base = m(x.cuda()).cpu()
        old = m.configure(plane_stride=((28, 800), (14, 224)))
        try:
            alt = m(x.cuda()).cpu()
        finally:
            m.configure(**old.__dict__)
        ref = OD.embed(x, sd)
    assert torch.equal(alt, base)
    assert float((alt - ref).abs().max()) <= 1e-5


@pytest.mark.gpu
def test_densenet_rows_do_not_depend_on_the_batch(model_and_sd):
    """The reference's eval-mode forward is strictly per row (model.py:71-84).  On the two-fp16-term path the value ranges
    are per image (one float per image and buffer), no kernel reduces across images, and a pixel tile that straddles two
    images scales each pixel by its own image's range -- so the same 6 images embedded alone and as rows of a 2048-image
    batch (other grid sizes, 64-bit strides, two pixel tiles per workgroup) must come out BIT-IDENTICAL, and within 1e-5 of
    the CPU restatement."""
    m, sd = model_and_sd
    g = torch.Generator(device="cuda").manual_seed(11)
    big = torch.randn(2048, 3, 224, 224, generator=g, device="cuda")
    big[5] *= 40.0                                     # a loud neighbour must not move anybody else's bits
    rows = [0, 1, 777, 1024, 2046, 2047]
    with torch.no_grad():
        e_big = m(big)[rows].cpu()
        e_small = m(big[rows].contiguous()).cpu()
        e_one = torch.cat([m(big[r:r + 1]) for r in rows]).cpu()
        ref = OD.embed(big[rows].cpu(), sd)
    assert torch.equal(e_big, e_small) and torch.equal(e_big, e_one)
    assert float((e_big - ref).abs().max()) <= 1e-5


@pytest.mark.gpu
def test_a_poisoned_image_leaves_its_batch_mates_unchanged(model_and_sd):
    """one image holding inf / NaN used to turn the batch-wide range, and with it every embedding of the batch,
    into NaN.  With per-image ranges the poisoned rows come out NaN (loud, never finite garbage) and every other row keeps
    exactly the bits it has in a clean batch."""
    m, _ = model_and_sd
    g = torch.Generator(device="cuda").manual_seed(12)
    x = torch.randn(9, 3, 224, 224, generator=g, device="cuda")
    with torch.no_grad():
        clean = m(x).cpu()
        bad = x.clone()
        bad[2, 1, 100, 100] = float("inf")
        bad[6, 0, 3, 200] = float("nan")
        out = m(bad).cpu()
    keep = [0, 1, 3, 4, 5, 7, 8]
    assert bool(torch.isnan(out[2]).all()) and bool(torch.isnan(out[6]).all())
    assert torch.equal(out[keep], clean[keep])


@pytest.mark.gpu
@pytest.mark.parametrize("side,batch,cin,mag,pad", [(56, 2, 64, 1.0, 0), (28, 3, 256, 25.0, 0), (14, 5, 512, 1e-2, 0),
                                                    (14, 1, 1008, 1.0, 0), (14, 3, 512, 1.0, 28), (28, 2, 128, 3.0, 16),
                                                    (56, 1, 96, 1.0, 32), (7, 5, 512, 1.0, 0), (7, 1, 992, 1.0, 0),
                                                    (7, 8, 640, 30.0, 0), (7, 3, 512, 1.0, 15)])
def test_dense_layer_terms_path_matches_float64(side, batch, cin, mag, pad, conv1x1_kernel):
    """conv1x1 -> pre-split fp16-term bottleneck -> conv3x3 (mirx_conv1x1_bn_relu_split2h_terms +
    mirx_conv3x3_direct_terms_nchw) against a float64 dense layer relu(bn2(conv1(relu(bn1(x))))) -> conv2: 3e-6 of the
    largest output at any input magnitude; the halo ring of the DMA-staged strips is zero (out-of-range buffer loads);
    neighbours of the written channel slice untouched; every image's output range published exactly; images of very
    different magnitudes in one batch (ranges are per image).  pad > 0: the block buffer's
    channel planes are `pad` floats apart beyond side^2 (plane stride); the gaps hold NaN, are never read and never written."""
    import ctypes
    from mirx import _lib
    from mirx.model import YTERMS_CHANNEL_ORDER, _conv3x3_weights_split2h, _split2h_weights
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(side + cin)
    ctot = cin + 64
    hw = side * side
    ps = hw + pad
    store = torch.full((batch, ctot, ps), float("nan"), device=dev)
    buf = store[:, :, :hw].unflatten(2, (side, side))                   # a view: [batch, ctot, side, side] with plane stride ps
    buf.copy_(torch.randn(batch, ctot, side, side, generator=g, device=dev) * mag)
    if batch > 1:
        buf[batch - 1] *= 1e-3                                          # a quiet image next to loud ones: its own scale
    buf[:, cin:] = 7.0 * mag
    sc = torch.rand(cin, generator=g, device=dev) + 0.5
    sh = torch.randn(cin, generator=g, device=dev) * 0.3 * mag
    w1 = torch.randn(128, cin, generator=g, device=dev) / cin ** 0.5
    b1 = torch.randn(128, generator=g, device=dev) * 0.2 * mag
    w3 = torch.randn(32, 128, 3, 3, generator=g, device=dev) * 0.05
    assert sorted(YTERMS_CHANNEL_ORDER) == list(range(128))
    w2, osc = _split2h_weights(w1)
    c3, c3osc = _conv3x3_weights_split2h(w3, YTERMS_CHANNEL_ORDER)
    y = torch.empty((batch, 128, side, side), device=dev)               # the same bytes, written as fp16 terms
    brange = buf[:, :cin].abs().amax(dim=(1, 2, 3)).contiguous()         # range row: one float per image
    yinv = torch.zeros(batch, device=dev)
    vp = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + off)      # noqa: E731
    _lib.check(lib.mirx_conv1x1_bn_relu_split2h_terms(vp(store), ctot * ps, cin, vp(sc), vp(sh), vp(w2), vp(osc), vp(b1), batch, hw,
                                                      vp(y), vp(brange), float(sc.abs().max()), float(sh.abs().max()),
                                                      float(w1.abs().sum(dim=1).max()), float(b1.abs().max()), vp(yinv),
                                                      ps if pad else 0, None), "terms")
    rng_before = brange.clone()
    _lib.check(lib.mirx_conv3x3_direct_terms_nchw(vp(y), vp(c3), vp(c3osc), batch, side, vp(store, 4 * cin * ps), ctot * ps, vp(yinv),
                                                  vp(brange), ps if pad else 0, None), "conv3x3_terms")
    torch.cuda.synchronize()
    x64 = torch.relu(buf[:, :cin].double().cpu() * sc.double().cpu()[None, :, None, None] + sh.double().cpu()[None, :, None, None])
    y64 = torch.relu(torch.einsum("oc,bchw->bohw", w1.double().cpu(), x64) + b1.double().cpu()[None, :, None, None])
    want = torch.nn.functional.conv2d(y64, w3.double().cpu(), None, padding=1)
    got = buf[:, cin:cin + 32].double().cpu()
    for b in range(batch):                                               # per image: the quiet one keeps its own accuracy
        assert float((got[b] - want[b]).abs().max()) < 3e-6 * float(want[b].abs().max()), b
    assert bool((buf[:, cin + 32:] == 7.0 * mag).all())                  # neighbours untouched
    assert pad == 0 or bool(torch.isnan(store[:, :, hw:]).all())         # plane gaps untouched (and, being NaN, unread)
    assert bool((yinv > 0).all())
    assert torch.equal(brange, torch.maximum(rng_before, buf[:, cin:cin + 32].abs().amax(dim=(1, 2, 3))))


@pytest.mark.gpu
@pytest.mark.parametrize("side,batch,cin,mag,pad", [(14, 1, 256, 1.0, 0), (14, 5, 512, 1e-2, 0), (14, 3, 992, 30.0, 0), (14, 2, 288, 1.0, 0),
                                                    (14, 2, 128, 1.0, 0), (14, 3, 160, 1.0, 0), (14, 2, 192, 1.0, 0), (14, 2, 224, 1.0, 0), (14, 2, 160, 1.0, 28),
                                                    (7, 1, 512, 1.0, 0), (7, 4, 544, 1.0, 0), (7, 9, 992, 25.0, 0), (7, 6, 640, 1.0, 0),
                                                    (14, 300, 320, 1.0, 0), (14, 515, 352, 1.0, 0), (7, 1027, 576, 1.0, 0)])
def test_fused_dense_layer_is_bit_identical_to_the_two_launches(side, batch, cin, mag, pad):
    """mirx_dense_layer_fused (14 x 14 / 7 x 7 maps: 1x1 conv -> bottleneck resident in LDS -> 3x3 conv, one persistent
    workgroup per CU) against mirx_conv1x1_bn_relu_split2h_terms + mirx_conv3x3_direct_terms_nchw on the same buffer: the 32
    new channels and every image's range are BIT-IDENTICAL (same MFMA sequences), the float64 dense layer is met to 3e-6 per
    image, neighbours and plane gaps untouched.  Batches that end inside a unit (7 x 7: four images per unit), more units
    than CUs (the persistent loop and its cross-unit prefetch), stage counts with nk % 4 == 2 (the ring rotation) and the
    shortest loops (nk = 8, 10, 12, 14: every tail of the unrolled stage loops), images of very different magnitudes."""
    import ctypes
    from mirx import _lib
    from mirx.model import YTERMS_CHANNEL_ORDER, _conv3x3_weights_split2h, _split2h_weights
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(side + cin + batch)
    ctot = cin + 64
    hw = side * side
    ps = hw + pad
    store = torch.full((batch, ctot, ps), float("nan"), device=dev)
    buf = store[:, :, :hw].unflatten(2, (side, side))
    buf.copy_(torch.randn(batch, ctot, side, side, generator=g, device=dev) * mag)
    if batch > 1:
        buf[batch - 1] *= 1e-3
    buf[:, cin:] = 7.0 * mag
    sc = torch.rand(cin, generator=g, device=dev) + 0.5
    sh = torch.randn(cin, generator=g, device=dev) * 0.3 * mag
    w1 = torch.randn(128, cin, generator=g, device=dev) / cin ** 0.5
    b1 = torch.randn(128, generator=g, device=dev) * 0.2 * mag
    w3 = torch.randn(32, 128, 3, 3, generator=g, device=dev) * 0.05
    w2, osc = _split2h_weights(w1)
    c3, c3osc = _conv3x3_weights_split2h(w3, YTERMS_CHANNEL_ORDER)
    brange0 = buf[:, :cin].abs().amax(dim=(1, 2, 3)).contiguous()
    ks, kb, yks, ykb = float(sc.abs().max()), float(sh.abs().max()), float(w1.abs().sum(dim=1).max()), float(b1.abs().max())
    vp = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + off)      # noqa: E731
    # two launches
    store_a, range_a = store.clone(), brange0.clone()
    y = torch.empty((batch, 128, side, side), device=dev)
    yinv = torch.zeros(batch, device=dev)
    _lib.check(lib.mirx_conv1x1_bn_relu_split2h_terms(vp(store_a), ctot * ps, cin, vp(sc), vp(sh), vp(w2), vp(osc), vp(b1), batch, hw,
                                                      vp(y), vp(range_a), ks, kb, yks, ykb, vp(yinv), ps if pad else 0, None), "terms")
    _lib.check(lib.mirx_conv3x3_direct_terms_nchw(vp(y), vp(c3), vp(c3osc), batch, side, vp(store_a, 4 * cin * ps), ctot * ps, vp(yinv),
                                                  vp(range_a), ps if pad else 0, None), "conv3x3_terms")
    # one launch
    store_b, range_b = store.clone(), brange0.clone()
    if pad:                                                            # padded planes are refused (stage offsets are immediates)
        assert lib.mirx_dense_layer_fused(vp(store_b), ctot * ps, ps, cin, vp(sc), vp(sh), vp(w2), vp(osc), vp(b1), vp(c3), vp(c3osc),
                                          batch, side, vp(range_b), ks, kb, yks, ykb, None) != 0
        return
    _lib.check(lib.mirx_dense_layer_fused(vp(store_b), ctot * ps, ps if pad else 0, cin, vp(sc), vp(sh), vp(w2), vp(osc), vp(b1), vp(c3),
                                          vp(c3osc), batch, side, vp(range_b), ks, kb, yks, ykb, None), "fused")
    torch.cuda.synchronize()
    a = store_a[:, :, :hw]
    b = store_b[:, :, :hw]
    assert torch.equal(a[:, cin:cin + 32], b[:, cin:cin + 32])
    assert torch.equal(range_a, range_b)
    assert torch.equal(b[:, :cin], store[:, :cin, :hw]) and bool((b[:, cin + 32:] == 7.0 * mag).all())     # prefix and neighbours untouched
    assert pad == 0 or bool(torch.isnan(store_b[:, :, hw:]).all())
    if batch <= 16:
        x64 = torch.relu(buf[:, :cin].double().cpu() * sc.double().cpu()[None, :, None, None] + sh.double().cpu()[None, :, None, None])
        y64 = torch.relu(torch.einsum("oc,bchw->bohw", w1.double().cpu(), x64) + b1.double().cpu()[None, :, None, None])
        want = torch.nn.functional.conv2d(y64, w3.double().cpu(), None, padding=1)
        got = b[:, cin:cin + 32].unflatten(2, (side, side)).double().cpu()
        for i in range(batch):
            assert float((got[i] - want[i]).abs().max()) < 3e-6 * float(want[i].abs().max()), i
    assert lib.mirx_dense_layer_fused(vp(store_b), ctot * ps, 0, cin, vp(sc), vp(sh), vp(w2), vp(osc), vp(b1), vp(c3), vp(c3osc), batch,
                                      28, vp(range_b), ks, kb, yks, ykb, None) != 0                      # 28 x 28 does not fit


@pytest.mark.gpu
def test_fused_and_two_launch_small_maps_give_the_same_embeddings(model_and_sd):
    m, sd = model_and_sd
    x = torch.randn(7, 3, 224, 224, generator=torch.Generator().manual_seed(21)).cuda()
    x[4] *= 20.0
    with torch.no_grad():
        two = m(x)
        old = m.configure(fused_small_maps=True)
        try:
            fused = m(x)
        finally:
            m.configure(**old.__dict__)
    assert torch.equal(fused, two)


@pytest.mark.gpu
def test_uint8_images_embed_bit_identically_to_the_normalised_tensor(model_and_sd):
    """forward() on raw uint8 [B, 3, 224, 224] (ToTensor + Normalize of test.py:1309-1332 applied inside the stem kernel
    through a 3 x 256 table) against forward() on the tensor the reference's CPU transform produces (u / 255, - mean, / std in
    fp32): the same bits, per-image ranges included; the CPU restatement is met to 1e-5."""
    m, sd = model_and_sd
    g = torch.Generator().manual_seed(31)
    u8 = torch.randint(0, 256, (6, 3, 224, 224), generator=g, dtype=torch.uint8)
    u8[3] = torch.randint(100, 140, (3, 224, 224), generator=g, dtype=torch.uint8)      # a flat image: its own, smaller range
    u8[4, :, :, :] = 0
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    xf = (u8.float() / 255.0 - mean) / std                     # torchvision ToTensor + Normalize, on the CPU like the reference
    assert torch.equal(m.normalize_uint8(u8), xf)              # CPU tensor in, CPU ops
    with torch.no_grad():
        e_u8 = m(u8.cuda()).cpu()
        e_f = m(xf.cuda()).cpu()
        ref = OD.embed(xf, sd)
    assert torch.equal(e_u8, e_f)
    assert float((e_u8 - ref).abs().max()) <= 1e-5
    with torch.no_grad():                                      # other sizes: normalised by torch ops, then the legacy path
        small = m(u8[:2, :, :160, :192].contiguous().cuda()).cpu()
        want = m(xf[:2, :, :160, :192].contiguous().cuda()).cpu()
    assert float((small - want).abs().max()) <= 1e-6
