"""SigLIP towers on libmirx (mirx.siglip) and the k_norm.hip kernels they stand on.  Reference statements: float64 torch
on the CPU; tolerances written per test."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

vp = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + off) if t is not None else None      # noqa: E731


@pytest.mark.parametrize("m,c,tpi", [(300, 768, 0), (5, 1152, 0), (1, 128, 0), (2 * 144, 128, 144), (3 * 50, 512, 50)])
def test_layernorm_kernel(m, c, tpi):
    from mirx import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(m + c)
    x = (torch.randn(m, c, generator=g) * 3 + 1.5).cuda()
    w, b = (torch.rand(c, generator=g) + 0.5).cuda(), torch.randn(c, generator=g).cuda()
    want = torch.nn.functional.layer_norm(x.double().cpu(), (c,), w.double().cpu(), b.double().cpu(), 1e-6)
    if tpi:
        y = torch.full((m // tpi, c, tpi), float("nan"), device="cuda")
        want = want.view(m // tpi, tpi, c).transpose(1, 2)
    else:
        y = torch.empty_like(x)
    _lib.check(lib.mirx_layernorm(vp(x), m, c, vp(w), vp(b), 1e-6, vp(y), tpi, None), "layernorm")
    torch.cuda.synchronize()
    assert float((y.double().cpu() - want).abs().max()) < 2e-6 * float(want.abs().max())
    if not tpi:                                               # in place, and without affine parameters
        plain = torch.nn.functional.layer_norm(x.double().cpu(), (c,), None, None, 1e-6)
        _lib.check(lib.mirx_layernorm(vp(x), m, c, None, None, 1e-6, vp(x), 0, None), "layernorm")
        torch.cuda.synchronize()
        assert float((x.double().cpu() - plain).abs().max()) < 2e-6 * float(plain.abs().max())
    assert lib.mirx_layernorm(vp(x), m, 6, None, None, 1e-6, vp(y), 0, None) == -1


@pytest.mark.parametrize("b,c,h,w,p,ln", [(2, 3, 56, 70, 14, False), (3, 3, 32, 32, 4, False), (2, 128, 24, 24, 2, True),
                                          (1, 256, 12, 12, 2, True), (2, 3, 30, 45, 14, False)])
def test_patchify_kernel(b, c, h, w, p, ln):
    """mirx_patchify_nchw rows times conv.weight.flatten(1) == the stride-p convolution (the contract that turns a patch
    embedding / downsample conv into a Linear), with the LayerNorm2d prologue, ragged edges and K padding."""
    from mirx import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(h * w + c)
    x = torch.randn(b, c, h, w, generator=g)
    gm, bt = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    k = c * p * p
    kpad = (k + 15) // 16 * 16
    gh, gw = h // p, w // p
    out = torch.full((b * gh * gw, kpad), float("nan"), device="cuda")
    xg, gg, bg = x.cuda(), gm.cuda(), bt.cuda()
    _lib.check(lib.mirx_patchify_nchw(vp(xg), b, c, h, w, p, vp(gg) if ln else None, vp(bg) if ln else None, 1e-6, vp(out),
                                      kpad, None), "patchify")
    torch.cuda.synchronize()
    xx = x.double()
    if ln:
        xx = torch.nn.functional.layer_norm(xx.permute(0, 2, 3, 1), (c,), gm.double(), bt.double(), 1e-6).permute(0, 3, 1, 2)
    want = torch.nn.functional.unfold(xx[:, :, :gh * p, :gw * p], kernel_size=p, stride=p)        # [b, c p p, gh gw]
    want = want.transpose(1, 2).reshape(b * gh * gw, k)
    got = out.double().cpu()
    assert float((got[:, :k] - want).abs().max()) < (3e-6 if ln else 0) + 1e-12
    assert bool((got[:, k:] == 0).all())
    conv = torch.nn.Conv2d(c, 8, p, p).double()
    y = (want @ conv.weight.flatten(1).t() + conv.bias).view(b, gh, gw, 8).permute(0, 3, 1, 2)
    assert float((y - conv(xx)).abs().max()) < 1e-10


@pytest.mark.parametrize("b,heads,dh,nq,nk,masked", [(3, 4, 16, 16, 16, True), (2, 16, 72, 64, 64, True), (5, 16, 72, 1, 1024, False),
                                                      (2, 2, 64, 7, 130, True), (1, 3, 32, 5, 5, False)])
def test_attention_small_kernel(b, heads, dh, nq, nk, masked):
    from mirx import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(nq * nk + dh)
    c = heads * dh
    q = torch.randn(b, nq, c, generator=g)
    kv = torch.randn(b, nk, 2 * c, generator=g)
    mask = None
    if masked:
        mask = torch.ones(b, nk, dtype=torch.uint8)
        for i in range(b):
            mask[i, max(1, nk - 3 - 5 * i):] = 0
    out = torch.full((b, nq, c), float("nan"), device="cuda")
    qg, kvg = q.cuda(), kv.cuda()
    mg = mask.cuda() if mask is not None else None
    _lib.check(lib.mirx_attention_small(vp(qg), c, vp(kvg), vp(kvg, 4 * c), 2 * c, vp(mg), b, heads, dh, nq, nk, dh ** -0.5,
                                        vp(out), None), "attention_small")
    torch.cuda.synchronize()
    qq = q.double().view(b, nq, heads, dh).transpose(1, 2)
    kk = kv[..., :c].double().view(b, nk, heads, dh).transpose(1, 2)
    vv = kv[..., c:].double().view(b, nk, heads, dh).transpose(1, 2)
    att = qq @ kk.transpose(2, 3) * dh ** -0.5
    if mask is not None:
        att = att.masked_fill(~mask.bool()[:, None, None, :], float("-inf"))
    want = (torch.softmax(att, -1) @ vv).transpose(1, 2).reshape(b, nq, c)
    assert float((out.double().cpu() - want).abs().max()) < 2e-6 * max(1.0, float(want.abs().max()))


def test_tiny_dual_encoder_gpu_matches_cpu_float64():
    """Both towers on the libmirx path (packed qkv, MFMA / small attention with the padding mask, tanh-GELU epilogue,
    LayerNorm and patch-gather kernels, pooling head) against the same modules in float64 on the CPU."""
    from mirx.siglip import SiglipDualEncoder
    torch.manual_seed(2)
    v = dict(hidden_size=144, intermediate_size=208, num_hidden_layers=2, num_attention_heads=2, image_size=84, patch_size=14)
    t = dict(hidden_size=144, intermediate_size=208, num_hidden_layers=2, num_attention_heads=2, vocab_size=300,
             max_position_embeddings=64, projection_size=144)
    m = SiglipDualEncoder(v, t).eval()
    ids = torch.randint(0, 300, (3, 64))
    mask = torch.ones(3, 64, dtype=torch.long)
    mask[0, 9:] = 0
    mask[1, 30:] = 0
    px = torch.randn(4, 3, 84, 84)
    with torch.no_grad():
        ref = m.double()
        rt, ri = ref.get_text_features(ids, mask), ref.get_image_features(px.double())
        ro = ref(input_ids=ids, pixel_values=px.double(), attention_mask=mask)
        g = m.float().cuda()
        gt, gi = g.get_text_features(ids.cuda(), mask.cuda()), g.get_image_features(px.cuda())
        go = g(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda())
    assert float((gt.double().cpu() - rt).abs().max()) < 1e-5 * max(1.0, float(rt.abs().max()))
    assert float((gi.double().cpu() - ri).abs().max()) < 1e-5 * max(1.0, float(ri.abs().max()))
    assert float((go.logits_per_image.double().cpu() - ro.logits_per_image).abs().max()) < 1e-4
    assert float((go.image_embeds.double().cpu() - ro.image_embeds).abs().max()) < 1e-5


def test_medsiglip_full_geometry_against_transformers_float64():
    """Config 5 at the REAL geometry (SigLIP so400m: 1152 wide, 27 layers, 16 heads of 72, 448 x 448 -> 1024 tokens,
    B = 4; text tower 27 layers, 64 tokens with padding): the libmirx path against transformers.SiglipModel built from a
    local config, run in float64 on the CPU with the same weights.  Tolerance 2e-5 of the largest feature (fp32-grade
    through 27 layers), unit-norm embeddings within 1e-5."""
    tr = pytest.importorskip("transformers")
    from mirx.siglip import MEDSIGLIP_TEXT, MEDSIGLIP_VISION, SiglipDualEncoder
    torch.manual_seed(3)
    m = SiglipDualEncoder().eval()
    with torch.no_grad():                                       # LayerNorm / bias statistics away from the init's 1 / 0
        for name, p in m.named_parameters():
            if name.endswith("norm.weight") or "layer_norm" in name and name.endswith("weight") or "layernorm.weight" in name:
                p.add_(0.1 * torch.randn_like(p))
            elif name.endswith("bias"):
                p.add_(0.02 * torch.randn_like(p))
    cfg = tr.SiglipConfig(text_config=dict(MEDSIGLIP_TEXT, bos_token_id=1, eos_token_id=2, pad_token_id=0),
                          vision_config=dict(MEDSIGLIP_VISION))
    cfg.text_config._attn_implementation = "eager"
    cfg.vision_config._attn_implementation = "eager"
    hf = tr.SiglipModel(cfg).eval()
    assert not set(m.state_dict()) ^ set(hf.state_dict())
    hf.load_state_dict(m.state_dict())
    hf = hf.double()
    px = torch.randn(4, 3, 448, 448)
    ids = torch.randint(0, 32000, (3, 64))
    mask = torch.ones(3, 64, dtype=torch.long)
    mask[0, 17:] = 0
    mask[2, 40:] = 0
    feat = lambda o: o if torch.is_tensor(o) else o.pooler_output      # noqa: E731
    with torch.no_grad():
        ri = feat(hf.get_image_features(pixel_values=px.double()))
        rt = feat(hf.get_text_features(input_ids=ids, attention_mask=mask))
        del hf
        g = m.cuda()
        gi = g.get_image_features(px.cuda()).double().cpu()
        gt = g.get_text_features(ids.cuda(), mask.cuda()).double().cpu()
    assert float((gi - ri).abs().max()) < 2e-5 * float(ri.abs().max())
    assert float((gt - rt).abs().max()) < 2e-5 * float(rt.abs().max())
    n = torch.nn.functional.normalize
    assert float((n(gi, dim=1) - n(ri, dim=1)).abs().max()) < 1e-5
    assert float((n(gt, dim=1) - n(rt, dim=1)).abs().max()) < 1e-5
    # zero-shot logits of the dual encoder: identical argmax, logits within 1e-3 at scale exp(logit_scale) = 10
    lo_ref = (n(ri, dim=1) @ n(rt, dim=1).t()) * float(m.logit_scale.exp()) + float(m.logit_bias)
    lo = g(input_ids=ids.cuda(), pixel_values=px.cuda(), attention_mask=mask.cuda()).logits_per_image.double().cpu()
    assert float((lo - lo_ref).abs().max()) < 1e-3 and torch.equal(lo.argmax(1), lo_ref.argmax(1))
