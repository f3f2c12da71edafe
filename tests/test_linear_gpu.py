"""mirx_linear_split3 (csrc/k_linear_s3.hip): the token-major Linear layer on three-term bf16 MFMA, against
a float64 restatement of  y = epi(x W^T + b).  Tolerance: 3e-6 of the largest |y| (the dropped cross
terms are <= 3 * 2^-24 per product; accumulation is fp32) -- the class of an fp32 GEMM."""
import ctypes
import dataclasses
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _vp(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _ref(x, w, b, act, res, gamma):
    v = x.double() @ w.double().t()
    if b is not None:
        v = v + b.double()
    if act:
        v = 0.5 * v * (1.0 + torch.erf(v / math.sqrt(2.0)))
    if res is not None:
        v = res.double() + (gamma.double() if gamma is not None else 1.0) * v
    return v


@pytest.mark.parametrize("m,k,n", [(1, 16, 128), (127, 768, 768), (300, 768, 2304), (1370 * 2 + 5, 3072, 768),
                                   (4096, 128, 512), (129, 1152, 4352), (200, 48, 128), (64, 80, 256), (33, 32, 128), (150, 1152, 4304), (70, 64, 10), (260, 4304, 1152)])
@pytest.mark.parametrize("act,use_res,use_gamma,use_bias", [(0, False, False, True), (1, False, False, True),
                                                            (0, True, True, True), (1, True, False, False)])
def test_linear_split3_matches_float64(m, k, n, act, use_res, use_gamma, use_bias):
    from mirx import _lib
    from mirx.model import _split3_weights
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(m * 7 + k + n + act)
    x = (torch.randn(m, k, generator=g) * 1.5).to(dev)
    w = (torch.randn(n, k, generator=g) / math.sqrt(k)).to(dev)
    b = torch.randn(n, generator=g).to(dev) if use_bias else None
    res = torch.randn(m, n, generator=g).to(dev) if use_res else None
    gamma = torch.randn(n, generator=g).to(dev) if use_gamma else None
    want = _ref(x, w, b, act, res, gamma)
    w3 = _split3_weights(w)
    y = res.clone() if use_res else torch.full((m, n), float("nan"), device=dev)     # in place over the residual
    _lib.check(lib.mirx_linear_split3(_vp(x), m, k, _vp(w3), _vp(b), n, act, _vp(y) if use_res else None, _vp(gamma),
                                      _vp(y), None), "mirx_linear_split3")
    torch.cuda.synchronize()
    err = float((y.double() - want).abs().max())
    assert err < 3e-6 * max(1.0, float(want.abs().max())), err


def test_linear_split3_argument_checks():
    from mirx import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    x = torch.zeros(4, 24, device=dev)
    y = torch.zeros(4, 128, device=dev)
    w = torch.zeros(128 * 24 * 3, dtype=torch.bfloat16, device=dev)
    assert lib.mirx_linear_split3(_vp(x), 4, 24, _vp(w), None, 128, 0, None, None, _vp(y), None) != 0   # k % 16
    assert lib.mirx_linear_split3(_vp(x), 4, 16, _vp(w), None, 128, 3, None, None, _vp(y), None) != 0   # act
    assert lib.mirx_linear_split3(_vp(x), 4, 16, _vp(w), None, 128, 2, _vp(y), None, _vp(y), None) != 0   # tanh-GELU has no residual form
    assert lib.mirx_linear_split3(_vp(x), 0, 16, _vp(w), None, 128, 0, None, None, _vp(y), None) == 0   # empty batch


def test_vit_block_split3_matches_rocblas_path():
    """The fused block path (4 split-3 Linear launches + attention kernel) against the module-by-module
    rocBLAS fp32 path of the same block."""
    import mirx.model as mm
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    blk = mm._VitBlock(768, 12).to(dev).eval()
    with torch.no_grad():
        blk.ls1.gamma.normal_()
        blk.ls2.gamma.normal_()
        for p in (blk.attn.qkv.bias, blk.attn.proj.bias, blk.mlp.fc1.bias, blk.mlp.fc2.bias):
            p.normal_(std=0.1)
        x = torch.randn(3, 257, 768, device=dev)
        got = blk(x)
        mm.set_kernel_config(blk, dataclasses.replace(mm.DEFAULT_CONFIG, linear_three_bf16=False))      # rocBLAS fp32 Linears
        try:
            want = blk(x)
        finally:
            mm.set_kernel_config(blk, mm.DEFAULT_CONFIG)
    assert float((got - want).abs().max()) < 2e-5 * float(want.abs().max())


@pytest.mark.parametrize("n_img,tpi,k,n,use_res", [(3, 144, 512, 128, True), (2, 577, 2048, 512, True),
                                                   (5, 36, 64, 256, False), (1, 9216, 512, 128, True)])
def test_linear_split3_nchw_matches_float64(n_img, tpi, k, n, use_res):
    """ConvNeXt block tail: channels-last input, NCHW output with the skip added."""
    from mirx import _lib
    from mirx.model import _split3_weights
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(n_img + tpi + k)
    x = torch.randn(n_img * tpi, k, generator=g).to(dev)
    w = (torch.randn(n, k, generator=g) / math.sqrt(k)).to(dev)
    b = torch.randn(n, generator=g).to(dev)
    res = torch.randn(n_img, n, tpi, generator=g).to(dev) if use_res else None
    want = (x.double() @ w.double().t() + b.double()).reshape(n_img, tpi, n).permute(0, 2, 1)
    if use_res:
        want = want + res.double()
    y = torch.full((n_img, n, tpi), float("nan"), device=dev)
    _lib.check(lib.mirx_linear_split3_nchw(_vp(x), n_img, tpi, k, _vp(_split3_weights(w)), _vp(b), n, _vp(res), None,
                                           _vp(y), None), "mirx_linear_split3_nchw")
    torch.cuda.synchronize()
    err = float((y.double() - want).abs().max())
    assert err < 3e-6 * max(1.0, float(want.abs().max())), err
    # with a per-(image, feature) input scale (the GRN factor folded into the staging)
    sc = (0.5 + torch.rand(n_img, k, generator=g)).to(dev)
    xs = (x.double().reshape(n_img, tpi, k) * sc.double()[:, None, :]).reshape(n_img * tpi, k)
    want = (xs @ w.double().t() + b.double()).reshape(n_img, tpi, n).permute(0, 2, 1)
    if use_res:
        want = want + res.double()
    y = torch.full((n_img, n, tpi), float("nan"), device=dev)
    _lib.check(lib.mirx_linear_split3_nchw(_vp(x), n_img, tpi, k, _vp(_split3_weights(w)), _vp(b), n, _vp(res), _vp(sc),
                                           _vp(y), None), "mirx_linear_split3_nchw")
    torch.cuda.synchronize()
    err = float((y.double() - want).abs().max())
    assert err < 3e-6 * max(1.0, float(want.abs().max())), err


@pytest.mark.parametrize("n_img,tpi,k,n,use_res", [(3, 144, 512, 128, True), (2, 577, 2048, 512, True),
                                                   (5, 36, 64, 256, False), (1, 9216, 512, 128, True)])
@pytest.mark.parametrize("xmax,loose", [(3.0, 1.0), (700.0, 64.0)])
def test_linear_split2h_nchw_matches_float64(n_img, tpi, k, n, use_res, xmax, loose):
    """mirx_linear_split2h_nchw (ConvNeXt block tail / downsample on two fp16 terms) against float64, without and with
    the per-(image, feature) input scale whose maximum the kernel reads from the device; `loose`: the caller's bound
    over-estimates max |x| by that factor (what a provable bound does) without costing accuracy."""
    import mirx.model as mm
    from mirx import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(n_img + tpi + k)
    x = (torch.randn(n_img * tpi, k, generator=g).clamp(-4, 4) / 4.0 * xmax).to(dev)
    lin = torch.nn.Linear(k, n).to(dev)
    with torch.no_grad():
        lin.bias.normal_()
    w2, ws = mm._linear_h2_weights(lin)
    w, b = lin.weight.detach(), lin.bias.detach()
    res = torch.randn(n_img, n, tpi, generator=g).to(dev) if use_res else None
    want = (x.double() @ w.double().t() + b.double()).reshape(n_img, tpi, n).permute(0, 2, 1)
    if use_res:
        want = want + res.double()
    y = torch.full((n_img, n, tpi), float("nan"), device=dev)
    _lib.check(lib.mirx_linear_split2h_nchw(_vp(x), n_img, tpi, k, _vp(w2), _vp(b), n, _vp(res), None, xmax * loose, None,
                                            1.0 / ws, _vp(y), None), "mirx_linear_split2h_nchw")
    torch.cuda.synchronize()
    err = float((y.double() - want).abs().max())
    assert err < 3e-6 * max(1.0, float(want.abs().max())), err
    # with the GRN factor folded into the staging; its maximum is device data
    sc = (0.25 + 3.0 * torch.rand(n_img, k, generator=g)).to(dev)
    smax = sc.abs().amax().reshape(1)
    xs = (x.double().reshape(n_img, tpi, k) * sc.double()[:, None, :]).reshape(n_img * tpi, k)
    want = (xs @ w.double().t() + b.double()).reshape(n_img, tpi, n).permute(0, 2, 1)
    if use_res:
        want = want + res.double()
    y = torch.full((n_img, n, tpi), float("nan"), device=dev)
    _lib.check(lib.mirx_linear_split2h_nchw(_vp(x), n_img, tpi, k, _vp(w2), _vp(b), n, _vp(res), _vp(sc), xmax * loose,
                                            _vp(smax), 1.0 / ws, _vp(y), None), "mirx_linear_split2h_nchw")
    torch.cuda.synchronize()
    err = float((y.double() - want).abs().max())
    assert err < 3e-6 * max(1.0, float(want.abs().max())), err
    # a non-finite device bound poisons the output; a scale without its bound is refused
    smax.fill_(float("inf"))
    _lib.check(lib.mirx_linear_split2h_nchw(_vp(x), n_img, tpi, k, _vp(w2), _vp(b), n, None, _vp(sc), xmax, _vp(smax), 1.0 / ws,
                                            _vp(y), None), "mirx_linear_split2h_nchw")
    torch.cuda.synchronize()
    assert bool(torch.isnan(y).all())
    assert lib.mirx_linear_split2h_nchw(_vp(x), n_img, tpi, k, _vp(w2), _vp(b), n, None, _vp(sc), xmax, None, 1.0 / ws, _vp(y),
                                        None) != 0


@pytest.mark.parametrize("n,hw,c", [(3, 144, 512), (2, 577, 192), (1, 5, 4), (4, 2304, 1024)])
def test_grn_norm_kernel_matches_torch(n, hw, c):
    from mirx import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(n + hw + c)
    x = torch.randn(n, hw, c, device=dev) * 2.0
    gx = torch.empty(n, c, device=dev)
    _lib.check(lib.mirx_grn_norm_nhwc(_vp(x), n, hw, c, _vp(gx), None), "mirx_grn_norm_nhwc")
    want = torch.linalg.vector_norm(x.double(), ord=2, dim=1)
    assert float(((gx.double() - want) / want).abs().max()) < 2e-6


@pytest.mark.parametrize("n,c", [(3, 512), (64, 4096), (1, 7), (5, 1031)])
def test_grn_scale_kernel_matches_torch(n, c):
    """mirx_grn_scale: 1 + weight * gx / (mean gx + eps) per image and the batch-wide max |scale| (the device-side bound of the
    two-fp16-term block tail), against the torch expression of timm's GlobalResponseNorm."""
    from mirx import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(n + c)
    gx = (torch.rand(n, c, generator=g) * 5).to(dev)
    w = torch.randn(c, generator=g).to(dev)
    scale = torch.full((n, c), float("nan"), device=dev)
    smax = torch.zeros(1, device=dev)                       # combined by atomic max: zeroed by the caller
    _lib.check(lib.mirx_grn_scale(_vp(gx), _vp(w), n, c, 1e-6, _vp(scale), _vp(smax), None), "mirx_grn_scale")
    torch.cuda.synchronize()
    want = 1.0 + w.double() * (gx.double() / (gx.double().mean(dim=-1, keepdim=True) + 1e-6))
    assert float((scale.double() - want).abs().max()) < 2e-6 * float(want.abs().max())
    assert float(smax) == float(scale.abs().max())


def test_convnext_block_fused_matches_module_path():
    import mirx.model as mm
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    blk = mm._CnxBlock(128).to(dev).eval()
    with torch.no_grad():
        blk.mlp.grn.weight.normal_()
        blk.mlp.grn.bias.normal_()
        x = torch.randn(3, 128, 24, 24, device=dev)
        got = blk(x)
        mm.set_kernel_config(blk, dataclasses.replace(mm.DEFAULT_CONFIG, linear_three_bf16=False))      # rocBLAS fp32 Linears
        try:
            want = blk(x)
        finally:
            mm.set_kernel_config(blk, mm.DEFAULT_CONFIG)
    assert got.shape == want.shape
    assert float((got - want).abs().max()) < 2e-5 * float(want.abs().max())


@pytest.mark.parametrize("m,k,n,act", [(300, 768, 2304, 0), (2745, 768, 3072, 1), (129, 1152, 4304, 0), (64, 48, 200, 1)])
@pytest.mark.parametrize("xmax", [3.0, 900.0])
def test_linear_split2h_matches_float64(m, k, n, act, xmax):
    """mirx_linear_split2h (two fp16 terms per operand, caller-supplied power-of-two scales) against float64: the
    tolerance of the bf16 three-term kernel.  Inputs bounded by `xmax` (what a LayerNorm bound provides)."""
    import mirx.model as mm
    from mirx import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(m + k + n)
    x = torch.randn(m, k, generator=g).clamp(-4, 4) / 4.0 * xmax
    x[0, 0] = xmax
    x = x.to(dev)
    lin = torch.nn.Linear(k, n).to(dev)
    with torch.no_grad():
        lin.weight.mul_(0.5)
        lin.bias.normal_()
    want = x.double() @ lin.weight.double().t() + lin.bias.double()
    if act:
        want = 0.5 * want * (1.0 + torch.erf(want / math.sqrt(2.0)))
    with torch.no_grad():
        got = mm._linear_h2(lin, x, xmax, act=act)
    torch.cuda.synchronize()
    err = float((got.double() - want).abs().max())
    assert torch.isfinite(got).all()
    assert err < 3e-6 * max(1.0, float(want.abs().max())), err


def test_layernorm_bound_holds():
    import mirx.model as mm
    torch.manual_seed(0)
    ln = torch.nn.LayerNorm(96)
    with torch.no_grad():
        ln.weight.normal_(1.0, 0.5)
        ln.bias.normal_(0.0, 0.3)
        x = torch.randn(4000, 96) * torch.logspace(-3, 3, 4000)[:, None]
        x[0] = 0.0
        x[0, 5] = 1e4                                               # one-hot row: the worst case of the bound
        assert float(ln(x).abs().max()) <= mm._layernorm_bound(ln) * (1 + 1e-6)


def test_linear_out_bound_holds():
    import mirx.model as mm
    torch.manual_seed(1)
    ln = torch.nn.LayerNorm(64)
    lin = torch.nn.Linear(64, 96)
    with torch.no_grad():
        ln.weight.normal_(1.0, 0.5)
        ln.bias.normal_(0.0, 0.3)
        lin.weight.normal_(0.0, 0.4)
        x = torch.randn(5000, 64) * torch.logspace(-2, 3, 5000)[:, None]
        y = lin(ln(x))
        assert float(y.abs().max()) <= mm._linear_out_bound(ln, lin) * (1 + 1e-6)
        assert float(y[:, 32:64].abs().max()) <= mm._linear_out_bound(ln, lin, slice(32, 64)) * (1 + 1e-6)
        assert mm._linear_out_bound(ln, lin, slice(32, 64)) <= mm._linear_out_bound(ln, lin) + 1e-9


@pytest.mark.parametrize("kind", ["split3", "split2h"])
def test_linear_tanh_gelu_epilogue(kind):
    """act = 2: the tanh-form GELU of the SigLIP MLP (transformers `gelu_pytorch_tanh`) fused in the Linear epilogue."""
    import mirx.model as mm
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    lin = torch.nn.Linear(144, 208).to(dev)
    x = torch.randn(3, 37, 144, device=dev).clamp_(-4, 4)
    with torch.no_grad():
        got = mm._linear_s3(lin, x, act=2) if kind == "split3" else mm._linear_h2(lin, x, 4.0, act=2)
        want = torch.nn.functional.gelu(lin.double()(x.double()), approximate="tanh")
    assert float((got.double() - want).abs().max()) < 3e-6 * max(1.0, float(want.abs().max()))


def _from_terms(t, k, scale):
    """terms rows (fp16 [m, ceil32(k) * 2]) back to float64 [m, k]"""
    m = t.shape[0]
    t = t.view(m, -1, 2, 32).double()
    return (t[:, :, 0] + t[:, :, 1]).reshape(m, -1)[:, :k] / scale


@pytest.mark.parametrize("m,k", [(1, 4), (37, 96), (300, 200), (513, 768), (64, 4304)])
def test_rows_to_terms_and_layernorm_terms(m, k):
    """mirx_rows_to_terms / mirx_layernorm_terms: hi + lo reproduces scale * x to 2^-21 of the bound (two fp16 terms), the
    padding features are zero, and the LayerNorm variant carries exactly the rows mirx_layernorm writes."""
    import mirx.model as mm
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(m + k)
    x = (torch.randn(m, k, generator=g) * 3.0).clamp(-20, 20).to(dev)
    xt, xs = mm._rows_to_terms(x, 20.0)
    torch.cuda.synchronize()
    kp = (k + 31) // 32 * 32
    assert xt.shape == (m, 2 * kp) and xs == 1024.0
    back = _from_terms(xt, kp, xs)
    assert float((back[:, :k] - x.double()).abs().max()) <= 20.0 * 2.0 ** -21
    assert float(back[:, k:].abs().max()) == 0.0 if kp > k else True
    # a strided view (every second row of a wider matrix) is read in place
    wide = (torch.randn(2 * m, k + 8, generator=g)).to(dev)
    if (k + 8) % 4 == 0:
        vt, vs = mm._rows_to_terms(wide[::2, :k], 8.0)
        assert float((_from_terms(vt, k, vs) - wide[::2, :k].double()).abs().max()) <= 8.0 * 2.0 ** -21
    if k % 4 == 0:
        ln = torch.nn.LayerNorm(k).to(dev)
        with torch.no_grad():
            ln.weight.normal_()
            ln.bias.normal_()
            bound = mm._layernorm_bound(ln)
            yt, ys = mm._layernorm_terms(ln, x, bound)
            want = mm._layernorm(ln, x)
        torch.cuda.synchronize()
        assert float((_from_terms(yt, k, ys) - want.double()).abs().max()) <= bound * 2.0 ** -21


@pytest.mark.parametrize("m,k,n", [(1, 32, 4), (300, 96, 200), (255, 200, 96), (257, 768, 2304), (1370 * 2 + 5, 768, 768),
                                   (513, 3072, 768), (260, 1152, 4304), (300, 4304, 1152), (4100, 64, 516), (5120, 96, 3328)])
@pytest.mark.parametrize("act,use_res,use_gamma,terms_out", [(0, False, False, False), (1, False, False, False),
                                                             (2, False, False, True), (0, True, True, False),
                                                             (0, True, False, False), (1, False, False, True)])
def test_linear_terms_matches_float64(m, k, n, act, use_res, use_gamma, terms_out):
    """mirx_linear_terms (csrc/k_linear_t2.hip: both operands pre-split, DMA-fed 256 x 256 tiles) against a float64
    restatement -- ragged token / output / feature counts, every epilogue, the terms-rows output; tile counts below the CU
    count (every tile cut along k), and 260 tiles (256 whole + 4 cut ones).  Tolerance class of mirx_linear_split2h, whose
    arithmetic it shares: 3e-6 of the largest |y|."""
    import mirx.model as mm
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(m * 3 + k + n + act)
    xmax = 6.0
    x = (torch.randn(m, k, generator=g) * 1.5).clamp(-xmax, xmax).to(dev)
    lin = torch.nn.Linear(k, n).to(dev)
    with torch.no_grad():
        lin.bias.normal_()
    res = torch.randn(m, n, generator=g).to(dev) if use_res else None
    gamma = torch.randn(n, generator=g).to(dev) if use_gamma else None
    want = x.double() @ lin.weight.double().t() + lin.bias.double()
    if act == 1:
        want = 0.5 * want * (1.0 + torch.erf(want / math.sqrt(2.0)))
    elif act == 2:
        want = torch.nn.functional.gelu(want, approximate="tanh")
    if use_res:
        want = res.double() + (gamma.double() if use_gamma else 1.0) * want
    want = want.detach()
    with torch.no_grad():
        xt, xs = mm._rows_to_terms(x, xmax)
        if terms_out:
            bound = float(want.abs().max()) * 1.01 + 1e-3
            yt, ys = mm._linear_terms(lin, xt, xs, (m,), act=act, terms_bound=bound)
            torch.cuda.synchronize()
            npad = (n + 31) // 32 * 32
            got = _from_terms(yt, npad, ys)
            assert float(got[:, n:].abs().max()) == 0.0 if npad > n else True      # the next Linear's padding features
            got = got[:, :n]
            tol = 3e-6 * max(1.0, float(want.abs().max())) + bound * 2.0 ** -21
        else:
            y = res.clone() if use_res else None
            got = mm._linear_terms(lin, xt, xs, (m,), act=act, res=y, gamma=gamma, out=y).double()     # in place over the residual
            torch.cuda.synchronize()
            tol = 3e-6 * max(1.0, float(want.abs().max()))
    assert torch.isfinite(got).all()
    err = float((got - want).abs().max())
    assert err < tol, (err, tol)


def test_linear_terms_argument_checks():
    from mirx import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    xt = torch.zeros(4, 64, dtype=torch.float16, device=dev)
    wt = torch.zeros(256, 64, dtype=torch.float16, device=dev)
    y = torch.zeros(4, 8, device=dev)
    f = ctypes.c_float
    ok = lambda *a: lib.mirx_linear_terms(*a) == 0       # noqa: E731
    assert ok(_vp(xt), 4, 32, _vp(wt), None, 8, 0, None, None, f(1.0), _vp(y), None, f(1.0), None, 0, None)
    assert not ok(_vp(xt), 4, 32, _vp(wt), None, 6, 0, None, None, f(1.0), _vp(y), None, f(1.0), None, 0, None)          # n % 4
    assert not ok(_vp(xt), 4, 32, _vp(wt), None, 8, 3, None, None, f(1.0), _vp(y), None, f(1.0), None, 0, None)          # act
    assert not ok(_vp(xt), 4, 32, _vp(wt), None, 8, 0, None, None, f(1.0), None, None, f(1.0), None, 0, None)            # no output
    assert not ok(_vp(xt), 4, 32, _vp(wt), None, 8, 0, None, None, f(1.0), _vp(y), _vp(xt), f(1.0), None, 0, None)       # two outputs
    assert not ok(_vp(xt), 4, 32, _vp(wt), None, 8, 1, _vp(y), None, f(1.0), _vp(y), None, f(1.0), None, 0, None)        # residual + activation
    assert not ok(_vp(xt), 4, 32, _vp(wt), None, 8, 0, None, _vp(y), f(1.0), _vp(y), None, f(1.0), None, 0, None)        # gamma without residual
    assert ok(_vp(xt), 0, 32, _vp(wt), None, 8, 0, None, None, f(1.0), _vp(y), None, f(1.0), None, 0, None)              # empty batch
    assert lib.mirx_rows_to_terms(_vp(y), 4, 8, 6, f(1.0), _vp(xt), None) != 0                                   # row_stride < k


def test_linear_terms_tail_split_agrees_with_whole_tiles():
    """The cut tiles of the last round (workspace + fix-up launch) against the same launch on whole tiles only."""
    import mirx.model as mm
    from mirx import _lib
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    m, k, n = 5120, 768, 3328                                    # 20 x 13 = 260 tiles: 256 whole, 4 cut into 24 pieces each
    assert _lib.load().mirx_linear_terms_workspace_bytes(m, k, n) == 4 * 24 * 256 * 256 * 4
    assert _lib.load().mirx_linear_terms_workspace_bytes(256 * 16, k, 256 * 16) == 0       # 256 tiles: nothing to cut
    lin = torch.nn.Linear(k, n).to(dev)
    x = torch.randn(m, k, device=dev).clamp_(-5, 5)
    with torch.no_grad():
        xt, xs = mm._rows_to_terms(x, 5.0)
        a = mm._linear_terms(lin, xt, xs, (m,), act=1)
        mm.set_kernel_config(lin, dataclasses.replace(mm.DEFAULT_CONFIG, linear_terms_split_tail=False))
        b = mm._linear_terms(lin, xt, xs, (m,), act=1)
        a2 = mm._linear_terms(lin, xt, xs, (m,), act=1)
    assert float((a - b).abs().max()) < 1e-6 * float(b.abs().max())
    assert torch.equal(b, a2)                                    # whole tiles: run to run identical
    # tiles are numbered in bands of 8 token tiles, output tile by output tile (k_linear_t2.hip, lt2_tile_mn): the last four
    # numbers are output tile 12 of the token tiles 16..19 of the last band -- every other tile is the same bits
    assert torch.equal(a[: 16 * 256], b[: 16 * 256])
    assert torch.equal(a[:, : 12 * 256], b[:, : 12 * 256])
    assert not torch.equal(a[16 * 256:, 12 * 256:], b[16 * 256:, 12 * 256:])      # (the cut tiles sum in another order)


def test_vit_block_terms_path_matches_split2h_path():
    """A ViT block on the DMA-fed Linear (LayerNorm -> terms rows, fc1 -> terms rows -> fc2) against the same block on
    mirx_linear_split2h: the same two-fp16-term arithmetic in another summation order."""
    import mirx.model as mm
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    blk = mm._VitBlock(768, 12).to(dev).eval()
    with torch.no_grad():
        blk.ls1.gamma.normal_()
        blk.ls2.gamma.normal_()
        for p in (blk.attn.qkv.bias, blk.attn.proj.bias, blk.mlp.fc1.bias, blk.mlp.fc2.bias):
            p.normal_(std=0.1)
        x = torch.randn(4, 1370, 768, device=dev)
        x0 = x.clone()
        assert mm._linear_terms_ok(blk, 4 * 1370, (blk.attn.qkv, blk.attn.proj, blk.mlp.fc1, blk.mlp.fc2), (1.0,))
        got = blk(x)
        assert torch.equal(x, x0)                                                    # the block does not write into its input
        mm.set_kernel_config(blk, dataclasses.replace(mm.DEFAULT_CONFIG, linear_terms_min_rows=0))
        try:
            want = blk(x)
        finally:
            mm.set_kernel_config(blk, mm.DEFAULT_CONFIG)
    assert float((got - want).abs().max()) < 3e-6 * float(want.abs().max())
