"""mirx_attention_qkv_f32 (fp32 MFMA flash attention on the packed qkv projection) against a float64
restatement of softmax(q k^T / sqrt(d)) v -- the attention of timm's vit_base_patch14_dinov2 blocks
(reference model.py:459-463).  Tolerance 3e-6 relative to the largest output (fp32 products and sums,
v_exp_f32; measured 2-3e-7 on unit-variance inputs)."""
import ctypes
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(qkv, heads, split3=False):
    from mirx import _lib
    lib = _lib.load()
    b, n, three, h, dh = qkv.shape
    out = torch.full((b, n, h * dh), float("nan"), dtype=torch.float32, device=qkv.device)
    st = ctypes.c_void_p(torch.cuda.current_stream(qkv.device).cuda_stream)
    fn = lib.mirx_attention_qkv_f32_split3 if split3 else lib.mirx_attention_qkv_f32
    _lib.check(fn(ctypes.c_void_p(qkv.data_ptr()), b, n, heads, dh, float(dh) ** -0.5,
                  ctypes.c_void_p(out.data_ptr()), st), "mirx_attention_qkv_f32")
    return out


def _ref(qkv):
    q, k, v = (qkv[:, :, i].double().permute(0, 2, 1, 3) for i in range(3))          # [b, h, n, dh]
    p = torch.softmax(q @ k.transpose(-1, -2) * q.shape[-1] ** -0.5, dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(qkv.shape[0], qkv.shape[1], -1)


@pytest.mark.parametrize("b,n,heads", [(1, 1, 1), (2, 5, 3), (1, 32, 2), (1, 33, 1), (3, 127, 2), (2, 128, 12),
                                       (1, 129, 1), (2, 257, 12), (1, 1370, 12)])
@pytest.mark.parametrize("dh", [64, 72])
def test_attention_matches_float64(b, n, heads, dh):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1000 * n + heads)
    qkv = torch.randn((b, n, 3, heads, dh), generator=g, device=dev) * 1.5
    out = _run(qkv, heads)
    ref = _ref(qkv)
    assert float((out.double() - ref).abs().max()) < 3e-6 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("b,n,heads", [(1, 1, 1), (2, 5, 3), (1, 32, 2), (1, 33, 1), (3, 127, 2), (2, 128, 12),
                                       (1, 129, 1), (2, 257, 12), (1, 1370, 12)])
@pytest.mark.parametrize("dh", [64, 72])
def test_attention_split3_matches_float64(b, n, heads, dh):
    """mirx_attention_qkv_f32_split3: both GEMMs on three-term bf16 MFMAs; same tolerance as the fp32 kernel, and
    a peaked case (logits ~ +-60).  head_dim 64 = the tuned kernel, 72 = the generic one."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1000 * n + heads + 7)
    qkv = torch.randn((b, n, 3, heads, dh), generator=g, device=dev) * 1.5
    out = _run(qkv, heads, split3=True)
    ref = _ref(qkv)
    assert float((out.double() - ref).abs().max()) < 3e-6 * max(1.0, float(ref.abs().max()))
    qkv[:, :, 0] *= 4.0
    qkv[:, n // 2, 1] *= 6.0
    out = _run(qkv, heads, split3=True)
    ref = _ref(qkv)
    assert torch.isfinite(out).all()
    # logits of +-50: the absolute error of a logit scales with sum |q_c k_c| (here ~2000): 3 * 2^-24 of that per
    # dropped cross term, i.e. a few 1e-6 relative on the probabilities
    assert float((out.double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("dh", [32, 96])
def test_attention_other_head_dims(dh):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(dh)
    qkv = torch.randn((2, 77, 3, 3, dh), generator=g, device=dev)
    ref = _ref(qkv)
    for split3 in (False, True):
        out = _run(qkv, 3, split3=split3)
        assert float((out.double() - ref).abs().max()) < 3e-6 * max(1.0, float(ref.abs().max()))


def test_attention_peaked_and_large_logits():
    """One key dominates (logits ~ +-60): the running maximum must keep exp2 in range."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(7)
    qkv = torch.randn((1, 200, 3, 2, 64), generator=g, device=dev)
    qkv[:, :, 0] *= 6.0
    qkv[:, 150, 1] *= 8.0
    out = _run(qkv, 2)
    assert torch.isfinite(out).all()
    ref = _ref(qkv)
    assert float((out.double() - ref).abs().max()) < 3e-6 * max(1.0, float(ref.abs().max()))


def test_attention_argument_errors():
    from mirx import _lib
    lib = _lib.load()
    x = torch.zeros((1, 4, 3, 1, 40), device="cuda:0")
    rc = lib.mirx_attention_qkv_f32(ctypes.c_void_p(x.data_ptr()), 1, 4, 1, 40, 0.1, ctypes.c_void_p(x.data_ptr()), None)
    assert rc == -1 and b"head_dim" in lib.mirx_last_error()          # MIRX_EINVAL
    assert lib.mirx_attention_qkv_f32(None, 0, 0, 1, 64, 0.1, None, None) == 0


def test_vit_block_uses_the_kernel_and_matches_sdpa():
    """The model path (mirx.model._VitAttention) against torch's own attention on the same weights."""
    from mirx.model import _VitAttention
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    att = _VitAttention(768, 12).eval().to(dev)
    x = torch.randn(2, 300, 768, device=dev)
    with torch.no_grad():
        got = att(x)
    with torch.enable_grad():                     # grad mode -> the plain torch path
        want = att(x).detach()
    assert float((got - want).abs().max()) < 1e-5


@pytest.mark.parametrize("b,n,heads", [(1, 1, 1), (2, 5, 3), (1, 33, 1), (3, 127, 2), (2, 257, 12), (1, 1370, 12)])
@pytest.mark.parametrize("amp,dh", [(1.5, 64), (40.0, 64), (1.5, 72), (40.0, 72), (3.0, 32), (3.0, 96)])
def test_attention_split2h_matches_float64(b, n, heads, amp, dh):
    """mirx_attention_qkv_f32_split2h: two fp16 terms per operand with caller-supplied bounds (here: the actual maxima,
    and values up to a few hundred) -- the tolerance of the other two kernels."""
    from mirx import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(100 * n + heads)
    qkv = torch.randn((b, n, 3, heads, dh), generator=g, device=dev) * 1.5
    qkv[:, :, 2] *= amp                                               # large values: the scales must absorb them
    out = torch.full((b, n, heads * dh), float("nan"), device=dev)
    bqk = float(qkv[:, :, :2].abs().max())
    bv = float(qkv[:, :, 2].abs().max())
    _lib.check(lib.mirx_attention_qkv_f32_split2h(ctypes.c_void_p(qkv.data_ptr()), b, n, heads, dh, dh ** -0.5, bqk, bv,
                                                  ctypes.c_void_p(out.data_ptr()), None), "mirx_attention_qkv_f32_split2h")
    torch.cuda.synchronize()
    ref = _ref(qkv)
    assert torch.isfinite(out).all()
    assert float((out.double() - ref).abs().max()) < 3e-6 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("b,n,heads,dh", [(2, 1370, 12, 64), (1, 1024, 16, 72), (3, 100, 6, 32), (2, 257, 12, 64)])
def test_attention_split2h_terms_output_carries_the_fp32_output(b, n, heads, dh):
    """mirx_attention_qkv_f32_split2h_terms: the same attention with the result written as terms rows (the input format of
    mirx_linear_terms).  hi + lo of every element reproduces the fp32 output of mirx_attention_qkv_f32_split2h to 2^-21 of
    the bound (two fp16 terms), padding features are zero."""
    from mirx import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(7 * n + heads)
    qkv = torch.randn((b, n, 3, heads, dh), generator=g, device=dev) * 1.5
    c = heads * dh
    cp = (c + 31) // 32 * 32
    out = torch.empty((b, n, c), device=dev)
    bqk, bv = float(qkv[:, :, :2].abs().max()), float(qkv[:, :, 2].abs().max())
    _lib.check(lib.mirx_attention_qkv_f32_split2h(ctypes.c_void_p(qkv.data_ptr()), b, n, heads, dh, dh ** -0.5, bqk, bv,
                                                  ctypes.c_void_p(out.data_ptr()), None), "mirx_attention_qkv_f32_split2h")
    scale = 2.0 ** math.floor(math.log2(32768.0 / bv))
    t = torch.full((b * n, 2 * cp), float("nan"), dtype=torch.float16, device=dev)
    _lib.check(lib.mirx_attention_qkv_f32_split2h_terms(ctypes.c_void_p(qkv.data_ptr()), b, n, heads, dh, dh ** -0.5, bqk, bv, scale,
                                                        ctypes.c_void_p(t.data_ptr()), None), "mirx_attention_qkv_f32_split2h_terms")
    torch.cuda.synchronize()
    tt = t.view(b * n, cp // 32, 2, 32).double()
    back = (tt[:, :, 0] + tt[:, :, 1]).reshape(b * n, cp) / scale
    if cp > c:
        assert torch.isnan(back[:, c:]).all()        # the launch writes features [0, c): padding belongs to the buffer's owner
    assert float((back[:, :c] - out.view(b * n, c).double()).abs().max()) <= bv * 2.0 ** -21
