"""Host-side pieces added in round 3 that need no GPU: the frozen per-model kernel configuration, the uint8 normalisation the
stem kernel's table reproduces, and the algorithmic-work formulas bench.py prices the other backbones with."""
import dataclasses
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kernel_config_is_frozen_per_model_and_reaches_every_submodule():
    import mirx.model as mm
    a, b = mm.DenseNet121(), mm.DenseNet121()
    assert a.kernel_config == mm.DEFAULT_CONFIG and not a.kernel_config.fused_small_maps
    old = a.configure(fused_small_maps=True, plane_stride=((14, 224),))
    assert old == mm.DEFAULT_CONFIG
    assert a.kernel_config.fused_small_maps and b.kernel_config == mm.DEFAULT_CONFIG          # another instance is untouched
    assert all(m._mirx_cfg is a.kernel_config for m in a.modules())
    with pytest.raises(dataclasses.FrozenInstanceError):
        a.kernel_config.fused_small_maps = False
    with pytest.raises(TypeError):
        a.configure(no_such_switch=True)
    a.configure(**old.__dict__)
    assert a.kernel_config == mm.DEFAULT_CONFIG
    assert not hasattr(mm, "SPLIT2H_DENSENET") and not hasattr(mm, "PLANE_STRIDE_H2")          # no module-level switches left
    blk = mm._VitBlock(64, 4)
    assert mm._cfg(blk.attn.qkv) is mm.DEFAULT_CONFIG                                            # a block used on its own: the defaults
    mm.set_kernel_config(blk, dataclasses.replace(mm.DEFAULT_CONFIG, linear_three_bf16=False))
    assert not mm._cfg(blk.mlp.fc1).linear_three_bf16


def test_uint8_input_is_the_reference_transform_on_the_cpu_path():
    """forward() on uint8 = forward() on (u / 255 - mean) / std, the fp32 operations of ToTensor + Normalize (test.py:1309-1332)."""
    import mirx.model as mm
    torch.manual_seed(0)
    m = mm.DenseNet121().eval()
    u8 = torch.randint(0, 256, (2, 3, 64, 64), dtype=torch.uint8, generator=torch.Generator().manual_seed(1))
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    xf = (u8.float() / 255.0 - mean) / std
    assert torch.equal(m.normalize_uint8(u8), xf)
    with torch.no_grad():
        assert torch.equal(m(u8), m(xf))
    assert "input_mean" not in m.state_dict()                   # non-persistent buffers: reference checkpoints load unchanged


def test_algorithmic_work_formulas_have_the_known_answers():
    """bench.py prices configs 3-5 with these: ConvNeXtV2-base at 384 x 384 is ~45 GMAC; a ViT block is
    24 T C^2 + 4 T^2 C FLOP."""
    sys.path.insert(0, ROOT)
    import bench
    flop, nbytes = bench.convnextv2_work(384)
    assert 44.5e9 < flop / 2 < 45.6e9
    assert 0.6e9 < nbytes < 0.75e9
    t, c = 1370, 768
    flop, nbytes = bench.vit_work(t, c, 4 * c, 12, 588)
    assert abs(flop - (2.0 * t * 588 * c + 12 * (24.0 * t * c * c + 4.0 * t * t * c))) < 1e6
    assert abs(nbytes - 4.0 * (t * (588 + c) + 12 * 20 * t * c)) < 1e3


def test_terms_rows_layout_and_the_tail_split_plan():
    """Host side of the token-major Linear on pre-split operands (include/mirx.h, mirx_linear_terms): the terms-rows layout of
    the weights (per 32 features one 128-byte line: fp16 high terms | fp16 low terms, zero padding) reproduces the scaled
    matrix to two fp16 terms; the launcher's plan (no GPU needed: it counts tiles) cuts only the last, partly filled round."""
    import mirx.model as mm
    from mirx import _lib
    torch.manual_seed(3)
    w = torch.randn(70, 200) * 3.0
    t = mm._terms_of(w, 8.0)                                            # [rows, ceil32(k) / 32, 2, 32] fp16
    assert t.shape == (70, 7, 2, 32) and t.dtype == torch.float16
    back = (t[:, :, 0].double() + t[:, :, 1].double()).reshape(70, 224) / 8.0
    assert float((back[:, :200] - w.double()).abs().max()) <= float(w.abs().max()) * 2.0 ** -21
    assert float(back[:, 200:].abs().max()) == 0.0
    lin = torch.nn.Linear(200, 70)
    wt, ws = mm._linear_terms_weights(lin)
    assert wt.shape == (256, 7, 2, 32) and float(wt[70:].abs().max()) == 0.0        # rows padded to the 256-output tile
    assert 2.0 ** 13 <= float(lin.weight.abs().max()) * ws < 2.0 ** 14
    assert mm._terms_scale(6.0) == 4096.0 and mm._terms_scale(32768.0) == 1.0
    lib = _lib.load()
    tile = 256 * 256 * 4
    assert lib.mirx_linear_terms_workspace_bytes(256 * 16, 768, 256 * 16) == 0                  # 256 tiles: whole rounds only
    assert lib.mirx_linear_terms_workspace_bytes(5120, 768, 3328) == 4 * 24 * tile             # 260 tiles: 4 cut into 24 pieces
    assert lib.mirx_linear_terms_workspace_bytes(300, 96, 200) == 2 * 3 * tile                 # 2 tiles, 3 stages: cut into 3
    assert lib.mirx_linear_terms_workspace_bytes(43840, 768, 768) == 4 * 24 * tile             # DINOv2 proj at 32 images: 516 tiles
    assert lib.mirx_linear_terms_workspace_bytes(256 * 200, 768, 256) == 0                     # 200 tiles: the round is 78 % full
    assert lib.mirx_linear_terms_workspace_bytes(0, 768, 768) == 0


def test_default_transform_is_the_float32_arithmetic_of_totensor_and_normalize():
    """retriever.default_transform computes (u / 255 - mean) / std in numpy float32 (torch's CPU operators cost 10+ ms per
    call on the GPU boxes); the result must be the bits the torch expression of ToTensor + Normalize gives."""
    import numpy as np
    from PIL import Image
    from mirx.retriever import IMAGENET_MEAN, IMAGENET_STD, default_transform
    img = Image.fromarray(np.random.default_rng(1).integers(0, 256, (300, 280, 3), dtype=np.uint8))
    got = default_transform(224)(img)
    w, h = img.size
    nw, nh = (256, int(256 * h / w)) if w <= h else (int(256 * w / h), 256)
    ref = img.convert("RGB").resize((nw, nh), Image.BILINEAR)
    left, top = int(round((nw - 224) / 2.0)), int(round((nh - 224) / 2.0))
    ref = ref.crop((left, top, left + 224, top + 224))
    x = torch.from_numpy(np.asarray(ref, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)
    want = (x - torch.tensor(IMAGENET_MEAN).view(3, 1, 1)) / torch.tensor(IMAGENET_STD).view(3, 1, 1)
    assert got.dtype == torch.float32 and got.shape == (3, 224, 224) and got.is_contiguous()
    assert torch.equal(got, want)


def test_channels_last_patch_rows_reproduce_the_stride2_convolution():
    """Host side of ConvNeXtV2's channels-last downsample: patch rows in (ky, kx, c) order (what mirx_layernorm_patch2_nhwc writes)
    times _ConvAsLinear(conv, channels_last=True).weight is the Conv2d(kernel = stride = 2) of the NCHW map."""
    import mirx.model as mm
    torch.manual_seed(5)
    conv = torch.nn.Conv2d(24, 40, kernel_size=2, stride=2)
    x = torch.randn(2, 24, 6, 8)
    want = conv(x).permute(0, 2, 3, 1).reshape(-1, 40)                                        # [b * 3 * 4, 40] rows
    t = x.permute(0, 2, 3, 1).contiguous()                                                    # NHWC
    rows = t.view(2, 3, 2, 4, 2, 24).permute(0, 1, 3, 2, 4, 5).reshape(2 * 3 * 4, 4 * 24)      # (ky, kx, c) per patch
    pl = mm._ConvAsLinear(conv, channels_last=True).refresh()
    assert pl.in_features == 96 and pl.out_features == 40
    got = rows @ pl.weight.t() + pl.bias
    assert float((got - want).abs().max()) < 1e-5
    plain = mm._ConvAsLinear(conv).refresh()                                                  # (c, ky, kx): the NCHW patch gather's order
    assert not torch.equal(plain.weight, pl.weight)


def test_tanh_gelu_sigmoid_form_is_as_accurate_as_the_tanh_form():
    """csrc/mirx_common.h:gelu_tanh computes v / (1 + 2^(v (K1 + K2 v^2))) -- the identity 1 + tanh(u) = 2 / (1 + e^(-2u)) -- with
    one v_exp_f32 and one v_rcp_f32.  Emulated in float32 with numpy: within 1e-6 (absolute) of the float64 value of the
    reference's activation (transformers "gelu_pytorch_tanh") and at least as accurate relatively as the tanh form itself."""
    import numpy as np
    f32 = np.float32
    t = np.linspace(-12.0, 12.0, 240001)
    v = t.astype(f32)
    k1 = f32(-2.302208185195923)                 # -2 sqrt(2/pi) log2(e): the constants of the kernel
    k2 = f32(-0.10294324159622192)               # 0.044715 K1
    assert abs(float(k1) + 2 * 0.7978845608028654 * 1.4426950408889634) < 1e-7
    assert abs(float(k2) - 0.044715 * float(k1)) < 1e-8
    with np.errstate(over="ignore"):
        e = np.exp2((v * ((v * v).astype(f32) * k2 + k1).astype(f32)).astype(f32)).astype(f32)
    got = (v * (f32(1) / (f32(1) + e).astype(f32)).astype(f32)).astype(f32)
    ref = 0.5 * t * (1 + np.tanh(0.7978845608028654 * (t + 0.044715 * t ** 3)))
    assert float(np.abs(got - ref).max()) < 1e-6
    u = (f32(0.7978845608028654) * (v + f32(0.044715) * v * v * v)).astype(f32)
    tanh_form = (f32(0.5) * v * (f32(1) + np.tanh(u.astype(np.float64)).astype(f32))).astype(f32)
    big = np.abs(ref) > 1e-3
    rel = lambda x: float((np.abs(x - ref)[big] / np.abs(ref[big])).max())
    assert rel(got) < 2e-6 and rel(got) <= rel(tanh_form)


def test_kernel_config_reaches_the_linear_views_that_are_not_modules():
    """a patch-embedding / downsample conv seen as a Linear (_ConvAsLinear) and SigLIP's packed q / k / v rows
    (_PackedRows) are plain objects, not nn.Modules -- configure() must reach them too, or A/B arms measured through it are mixed."""
    import mirx.model as mm
    from mirx.siglip import SiglipVisionTower
    torch.manual_seed(0)
    v = SiglipVisionTower(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2, image_size=28, patch_size=14)
    off = dataclasses.replace(mm.DEFAULT_CONFIG, linear_two_fp16=False)
    mm.set_kernel_config(v, off)
    packed = [m.__dict__["_packed"] for m in v.modules() if "_packed" in m.__dict__]
    assert packed and all(mm._cfg(p) is off for p in packed)
    conv = torch.nn.Conv2d(3, 8, 4, 4)
    mm.set_kernel_config(conv, off)
    view = mm._ConvAsLinear(conv)
    conv.__dict__["_mirx_as_linear"] = view
    mm.set_kernel_config(conv, off)
    assert mm._cfg(view) is off
