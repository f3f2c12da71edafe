"""Embedding models with the reference's forward() contract (model.py of the reference).

Contract kept (reference model.py:50-84): ``DenseNet121(pretrained, embedding_dim, num_labels)``,
``forward(x: float[B,3,H,W]) -> float[B,D]`` with unit-norm rows (eps 1e-12), a dict with
``"embedding"``/``"logits"`` when ``num_labels`` is set, attributes ``densenet121`` (Sequential
whose child ``0`` is the feature stack and whose child ``avgpool`` is the pooling), ``fc``,
``classification_head``; state-dict keys ``densenet121.0.conv0.weight``,
``densenet121.0.denseblock1.denselayer1.norm1.weight``, ..., ``densenet121.0.norm5.*``,
``fc.*`` load unchanged.

The backbone is restated here from the published DenseNet-121 definition (Huang et al.;
torchvision is NOT a dependency and is absent offline): conv0 7x7/2 (3->64), BN, ReLU,
max-pool 3x3/2, dense blocks of (6, 12, 24, 16) layers [BN-ReLU-conv1x1(->128)-BN-ReLU-
conv3x3(->32)], transitions [BN-ReLU-conv1x1(halve)-avgpool 2x2], norm5.  6 953 856
feature parameters, final map [B,1024,H/32,W/32].

MI355X path (CUDA tensors, eval mode):
  * stem conv0+norm0+relu0+pool0 -> one HIP kernel (libmirx ``mirx_stem_conv7_bn_relu_pool``)
  * dense blocks: convolutions run under PyTorch-ROCm (MIOpen / rocBLAS, fp32); each block owns
    ONE preallocated feature buffer (no O(L^2) torch.cat re-copies); norm1+relu1 is one HIP pass
    over the buffer's channel prefix (``mirx_bn_relu_nchw``); norm2 is folded into conv1
  * transitions: norm+relu+avgpool in one HIP pass (``mirx_bn_relu_avgpool2``), then the 1x1
    conv on the pooled map (conv and average pool commute)
  * norm5+relu+global-average-pool+flatten+F.normalize -> one HIP kernel
    (``mirx_bn_relu_gap_l2norm``), or GAP only when an ``fc`` follows.
``pretrained=True`` would need a download in the reference (model.py:53); here it raises
unless a local state dict is given via ``weights=``.
"""
import ctypes
import dataclasses
import math
import operator
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

IMAGENET_MEAN = (0.485, 0.456, 0.406)          # reference test.py:1309-1310
IMAGENET_STD = (0.229, 0.224, 0.225)
BLOCK_CONFIG = (6, 12, 24, 16)
GROWTH = 32
BN_SIZE = 4
INIT_FEATURES = 64


class _DenseLayer(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(cin)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(cin, BN_SIZE * GROWTH, kernel_size=1, stride=1, bias=False)
        self.norm2 = nn.BatchNorm2d(BN_SIZE * GROWTH)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(BN_SIZE * GROWTH, GROWTH, kernel_size=3, stride=1, padding=1, bias=False)

    def forward(self, x):
        y = self.conv1(self.relu1(self.norm1(x)))
        return self.conv2(self.relu2(self.norm2(y)))


class _DenseBlock(nn.ModuleDict):
    def __init__(self, nlayers, cin):
        super().__init__()
        self.cin = cin
        for i in range(nlayers):
            self.add_module(f"denselayer{i + 1}", _DenseLayer(cin + i * GROWTH))

    @property
    def cout(self):
        return self.cin + len(self) * GROWTH

    def forward(self, x):
        # one buffer for the whole block: layer i reads channels [0, cin + 32 i) and appends 32
        b, _, h, w = x.shape
        buf = x.new_empty((b, self.cout, h, w))
        buf[:, : self.cin] = x
        c = self.cin
        for layer in self.values():
            buf[:, c: c + GROWTH] = layer(buf[:, :c])
            c += GROWTH
        return buf


@dataclasses.dataclass(frozen=True)
class KernelConfig:
    """Which kernel family each piece of the inference path runs on.  FROZEN: a model holds one instance (`model.kernel_config`,
    every submodule carries a reference) and `model.configure(name=value, ..)` swaps in a modified copy -- no module-level
    switch, nothing another thread or another model instance can observe half-changed.  The defaults are the shipped path; the
    rest are measured alternatives kept for A/B runs and for inputs the default path does not cover."""
    densenet_two_fp16: bool = True          # DenseNet at 224 x 224: every conv on two fp16 terms, per-image value ranges
    plane_stride: tuple = ()                # ((side, floats between channel planes), ..): padded block buffers (measured: no gain)
    fused_small_maps: bool = False          # 14 x 14 / 7 x 7 dense layers in ONE launch (mirx_dense_layer_fused; slower so far)
    pooled_twin: bool = True                # DenseNet: the transition's norm + relu + avgpool of a layer's new channels written by its 3x3 launch
    hip_stem: bool = True                   # legacy (not 224 x 224) path: the stem kernel (False: torch ops)
    stem_three_bf16: bool = True            #   .. on three bf16 terms (False: fp32 MFMAs)
    hip_conv1x1: bool = True                #   fused 1x1 convs (False: rocBLAS via torch)
    conv1x1_three_bf16_min_cin: int = 0     #   1x1 convs with at least this many input channels on three bf16 terms
    hip_conv3x3: bool = True                #   own 3x3 convs on the 56 / 28 / 14 / 7 maps (False: MIOpen)
    conv3x3_legacy: tuple = ((56, "wino"), (28, "wino3"), (14, "direct3"), (7, "wino"))
    linear_three_bf16: bool = True          # token-major Linears on the MFMA kernels at all (False: rocBLAS fp32)
    linear_two_fp16: bool = True            #   .. on two fp16 terms where the input has a provable bound (LayerNorm outputs)
    cnx_grn_in_fc1: bool = True             #   .. and fc1's epilogue returns the GRN norm's partial sums (no norm pass over the hidden map)
    cnx_channels_last: bool = True          # ConvNeXtV2: the residual stream stays NHWC between blocks (no-LDS depthwise conv, row-major block tail)
    linear_terms_split_tail: bool = True    #   .. its last, partly filled round of tiles cut along K (mirx.h, mirx_linear_terms workspace)
    linear_terms_min_rows: int = 1024       #   .. ViT / SigLIP blocks with at least this many token rows: the DMA-fed Linear on pre-split
                                            #      "terms rows" (k_linear_t2: 256 x 256 tiles); 0 = never
    attention_three_bf16: bool = True       # flash attention on three bf16 terms (False: fp32 MFMAs)
    attention_two_fp16: bool = True         #   .. on two fp16 terms where q / k / v have provable bounds
    grn_scale_kernel: bool = True           # ConvNeXtV2: GRN scale vector + its maximum in one launch (False: five ATen launches)


DEFAULT_CONFIG = KernelConfig()


def _cfg(mod):
    """The kernel configuration a module runs under (its model's, or the defaults for a module used on its own)."""
    return getattr(mod, "_mirx_cfg", DEFAULT_CONFIG)


def set_kernel_config(module, cfg):
    """Put every submodule of `module` under `cfg` (what model.configure() does; also for a block used on its own)."""
    assert isinstance(cfg, KernelConfig)
    for m in module.modules():
        m.__dict__["_mirx_cfg"] = cfg
        # the duck-typed Linear views that are not nn.Modules (a patch-embedding / downsample conv seen as a Linear, SigLIP's
        # packed q / k / v rows): they run under their owner's configuration too
        for key in ("_mirx_as_linear", "_mirx_as_linear_cl", "_packed"):
            view = m.__dict__.get(key)
            if view is not None:
                view.__dict__["_mirx_cfg"] = cfg


class _Configurable:
    """Mixin of the top-level models: `kernel_config` and `configure(**changes)`."""

    def _init_config(self):
        set_kernel_config(self, DEFAULT_CONFIG)

    def _set_config(self, cfg):
        set_kernel_config(self, cfg)

    @property
    def kernel_config(self):
        return _cfg(self)

    def configure(self, **changes):
        """Swap in a copy of the configuration with `changes` applied; returns the previous configuration (to restore)."""
        old = _cfg(self)
        self._set_config(dataclasses.replace(old, **changes))
        return old


def _timer_start(timer):
    """Optional per-launch timing of the fused conv kernel (bench.py's calibration pass)."""
    if timer is None:
        return None
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev[0].record()
    return ev


def _timer_stop(timer, ev, flops, nbytes=0.0, kind="conv1x1"):
    """(start event, stop event, FLOP, algorithmic bytes, kernel family) per launch."""
    if timer is not None:
        ev[1].record()
        timer.append((ev[0], ev[1], flops, nbytes, kind))


def _fused_bn_relu(lib, buf, c, scale, shift):
    """relu(bn(buf[:, :c])) -> packed [B, c, H, W] in one HIP pass over the channel-prefix view."""
    b, ctot, h, w = buf.shape
    out = torch.empty((b, c, h, w), dtype=torch.float32, device=buf.device)
    _lib.check(lib.mirx_bn_relu_nchw(_ptr(buf), ctot * h * w, _ptr(scale), _ptr(shift), b, c, h * w,
                                     _ptr(out), _stream(buf.device)), "mirx_bn_relu_nchw")
    return out


def _dense_block_fused(block, x, cache, cfg, timer=None):
    """Inference path of one dense block (CUDA, eval):
       per layer: [HIP] norm1+relu1 over the buffer prefix -> conv1 (1x1, norm2's scale folded into
       its weights) -> [HIP] norm2 shift + relu2 in place -> conv2 (3x3) -> 32 new channels copied
       into the block buffer."""
    lib = _lib.load()
    if isinstance(x, tuple):           # (block buffer, channels already in place): the transition wrote them there
        buf = x[0]
        x = buf
        b, _, h, w = buf.shape
    else:
        b, _, h, w = x.shape
        buf = torch.empty((b, block.cout, h, w), dtype=torch.float32, device=x.device)
        buf[:, : block.cin] = x
    c = block.cin
    for name, layer in block.items():
        sc1, sh1, w1, b1, ones, w1t, u3, w3 = cache[name]
        if cfg.hip_conv1x1:
            # norm1 + relu1 + conv1 + norm2 + relu2 in ONE fp32-MFMA pass over the buffer prefix
            y = torch.empty((b, w1t.shape[1], h, w), dtype=torch.float32, device=x.device)
            ev = _timer_start(timer)
            if c >= cfg.conv1x1_three_bf16_min_cin:
                # three-bf16-term MFMA formulation (fp32-grade, faster than the fp32-MFMA kernel on every layer)
                _lib.check(lib.mirx_conv1x1_bn_relu_split3(_ptr(buf), block.cout * h * w, c, _ptr(sc1), _ptr(sh1),
                                                           _ptr(w3), _ptr(b1), b, h * w, w1t.shape[1], 1, _ptr(y),
                                                           w1t.shape[1] * h * w, _stream(x.device)),
                           "mirx_conv1x1_bn_relu_split3")
            else:
                _lib.check(lib.mirx_conv1x1_bn_relu(_ptr(buf), block.cout * h * w, c, _ptr(sc1), _ptr(sh1), _ptr(w1t),
                                                    _ptr(b1), b, h * w, w1t.shape[1], 1, _ptr(y), _stream(x.device)),
                           "mirx_conv1x1_bn_relu")
            _timer_stop(timer, ev, 2.0 * b * h * w * c * w1t.shape[1])
            if cfg.hip_conv3x3 and h == w and h in (56, 28, 14, 7) and b <= 65535:
                # written straight into this layer's slice of the buffer.  Per map side (measured, 1024 images):
                # 56: Winograd F(2x2,3x3) on fp32 MFMAs (1.37 ms; the bf16 variants are operand-delivery bound there);
                # 28: Winograd on three-term bf16 MFMAs (0.32 vs 0.35 ms); 14: direct implicit GEMM on three-term
                # bf16 MFMAs (0.078 vs 0.090 / 0.097 ms); 7: Winograd fp32, two images per workgroup
                dst = ctypes.c_void_p(buf.data_ptr() + 4 * c * h * w)
                kind = dict(cfg.conv3x3_legacy).get(h, "wino")
                if kind == "direct3":
                    _lib.check(lib.mirx_conv3x3_direct_split3_nchw(_ptr(y), _ptr(u3[2]), b, h, dst, block.cout * h * w,
                                                                   _stream(x.device)), "mirx_conv3x3_direct_split3_nchw")
                elif kind == "wino3":
                    _lib.check(lib.mirx_conv3x3_winograd_split3_nchw(_ptr(y), _ptr(u3[1]), b, h, dst, block.cout * h * w,
                                                                     _stream(x.device)),
                               "mirx_conv3x3_winograd_split3_nchw")
                else:
                    _lib.check(lib.mirx_conv3x3_winograd_nchw(_ptr(y), _ptr(u3[0]), b, h, dst, block.cout * h * w,
                                                              _stream(x.device)), "mirx_conv3x3_winograd_nchw")
            else:
                buf[:, c: c + GROWTH] = F.conv2d(y, layer.conv2.weight, None, padding=1)
            c += GROWTH
            continue
        y = _fused_bn_relu(lib, buf, c, sc1, sh1)
        y = F.conv2d(y, w1)                       # norm2's scale is folded into w1 ...
        # ... and its shift + relu2 run as ONE in-place HIP pass (a conv bias would cost a separate
        # add kernel plus a clamp kernel under PyTorch-ROCm)
        _lib.check(lib.mirx_bn_relu_nchw(_ptr(y), y.shape[1] * h * w, _ptr(ones), _ptr(b1), b, y.shape[1], h * w,
                                         _ptr(y), _stream(y.device)), "mirx_bn_relu_nchw")
        y = F.conv2d(y, layer.conv2.weight, None, padding=1)
        buf[:, c: c + GROWTH] = y
        c += GROWTH
    return buf


def _transition_fused(tr, buf, cache, cfg, timer=None, next_channels=None):
    """[HIP] norm+relu+avgpool2x2 in one pass, then the 1x1 conv on the POOLED map (the conv and
    the average pool are both linear and commute; 4x fewer pixels go through the conv)."""
    lib = _lib.load()
    sc, sh, wt, w3 = cache
    b, c, h, w = buf.shape
    if h % 2 or w % 2:
        return tr.pool(tr.conv(F.relu(F.batch_norm(buf, tr.norm.running_mean, tr.norm.running_var,
                                                   tr.norm.weight, tr.norm.bias, False, 0.0, tr.norm.eps))))
    pooled = torch.empty((b, c, h // 2, w // 2), dtype=torch.float32, device=buf.device)
    _lib.check(lib.mirx_bn_relu_avgpool2(_ptr(buf), c * h * w, _ptr(sc), _ptr(sh), b, c, h, w, _ptr(pooled), 0,
                                         _stream(buf.device)), "mirx_bn_relu_avgpool2")
    if cfg.hip_conv1x1 and wt.shape[1] % 128 == 0 and c % 32 == 0:
        ev = _timer_start(timer)
        if c >= cfg.conv1x1_three_bf16_min_cin:
            # written straight into the channel prefix of the NEXT dense block's buffer when its width is known
            ctot = next_channels if next_channels else wt.shape[1]
            out = torch.empty((b, ctot, h // 2, w // 2), dtype=torch.float32, device=buf.device)
            _lib.check(lib.mirx_conv1x1_bn_relu_split3(_ptr(pooled), c * (h // 2) * (w // 2), c, None, None, _ptr(w3), None,
                                                       b, (h // 2) * (w // 2), wt.shape[1], 0, _ptr(out),
                                                       ctot * (h // 2) * (w // 2), _stream(buf.device)),
                       "mirx_conv1x1_bn_relu_split3")
            _timer_stop(timer, ev, 2.0 * b * (h // 2) * (w // 2) * c * wt.shape[1])
            return (out, wt.shape[1]) if next_channels else out
        else:
            out = torch.empty((b, wt.shape[1], h // 2, w // 2), dtype=torch.float32, device=buf.device)
            _lib.check(lib.mirx_conv1x1_bn_relu(_ptr(pooled), c * (h // 2) * (w // 2), c, None, None, _ptr(wt), None, b,
                                                (h // 2) * (w // 2), wt.shape[1], 0, _ptr(out), _stream(buf.device)),
                       "mirx_conv1x1_bn_relu")
        _timer_stop(timer, ev, 2.0 * b * (h // 2) * (w // 2) * c * wt.shape[1])
        return out
    return F.conv2d(pooled, tr.conv.weight)


def _plane_stride(side, cfg=DEFAULT_CONFIG):
    """Floats between consecutive channel planes of a dense block's buffer on the two-fp16-term path.
    784-byte (14 x 14) and 3136-byte (28 x 28) planes start at every 16-byte offset of a 128-byte line, so a wave's 256-byte
    load straddles three lines instead of two; in isolation the block-3 conv1x1 layers run at 4.1 instead of 3.5 TB/s on
    line-aligned planes (stride 224).  In the whole forward the padded layouts measured 1 % SLOWER (42.4 vs 42.8 k img/s,
    same box, B = 4096), so the default stays packed; the kernels and the ABI take the stride
    (KernelConfig.plane_stride, tools/bench_embed.py --plane-stride 14:224)."""
    hw = side * side
    ps = dict(cfg.plane_stride).get(side, hw)
    assert ps == hw or (ps > hw and ps % 4 == 0)
    return ps


def _dense_block_h2(block, buf, side, brange, cache, lranges, timer=None, fused_small=False, pool=None):
    """One dense block on the two-fp16-term kernels.  `buf` [B, block.cout, plane stride] (side x side pixels per plane, see
    _plane_stride) already holds the first block.cin channels and `brange` (range row [B]: one float per image) bounds them.
    56 / 28 maps, every layer: conv1x1 (norm1 + relu1 prologue, norm2 + relu2 epilogue) writes the bottleneck ALREADY SPLIT
    into fp16 terms, image b scaled by a bound derived from brange[b] before the kernel runs (2^-t goes to the layer's row of
    `lranges`); conv3x3 stages those terms by DMA, writes 32 channels into the buffer and folds their range into brange.
    14 / 7 maps with `fused_small`: ONE launch per layer, mirx_dense_layer_fused -- the bottleneck of a 196-pixel unit stays in
    the CU's LDS; bit-identical to the two launches and, so far, slower than them, hence opt-in.
    `pool` = (scale, shift, pooled) of the transition behind the block: every 3x3 launch then also writes norm + relu + avgpool2
    of its 32 new channels into `pooled` [B, block.cout, side/2, side/2] (mirx_conv3x3_direct_terms_nchw_pool; the caller has
    checked that the launches take the strip kernel)."""
    lib = _lib.load()
    b, _, ps = buf.shape
    h = w = side
    st = _stream(buf.device)
    c = block.cin
    fused = fused_small and side in (14, 7) and ps == h * w          # the fused kernel wants packed planes
    y = None if fused else torch.empty((b, BN_SIZE * GROWTH, h, w), dtype=torch.float32, device=buf.device)   # fp32 bytes, as terms
    for li, name in enumerate(block.keys()):
        e = cache[name]
        ev = _timer_start(timer)
        if fused:
            _lib.check(lib.mirx_dense_layer_fused(_ptr(buf), block.cout * ps, ps, c, _ptr(e["sc1"]), _ptr(e["sh1"]), _ptr(e["w2"]),
                                                  _ptr(e["osc"]), _ptr(e["b1"]), _ptr(e["c3w2p"]), _ptr(e["c3osc"]), b, side,
                                                  _ptr(brange), e["ks"], e["kb"], e["yks"], e["ykb"], st),
                       "mirx_dense_layer_fused")
            _timer_stop(timer, ev, 2.0 * b * h * w * (c * BN_SIZE * GROWTH + 9 * BN_SIZE * GROWTH * GROWTH),
                        4.0 * b * h * w * (c + GROWTH), "dense_fused")
            c += GROWTH
            continue
        dst = ctypes.c_void_p(buf.data_ptr() + 4 * c * ps)
        _lib.check(lib.mirx_conv1x1_bn_relu_split2h_terms(_ptr(buf), block.cout * ps, c, _ptr(e["sc1"]), _ptr(e["sh1"]),
                                                          _ptr(e["w2"]), _ptr(e["osc"]), _ptr(e["b1"]), b, h * w, _ptr(y),
                                                          _ptr(brange), e["ks"], e["kb"], e["yks"], e["ykb"],
                                                          _ptr(lranges[li]), ps, st), "mirx_conv1x1_bn_relu_split2h_terms")
        _timer_stop(timer, ev, 2.0 * b * h * w * c * y.shape[1], 4.0 * b * h * w * (c + y.shape[1]), "conv1x1")
        if pool is not None and not fused:
            psc, psh, pooled = pool
            hw2 = (h // 2) * (w // 2)
            _lib.check(lib.mirx_conv3x3_direct_terms_nchw_pool(_ptr(y), _ptr(e["c3w2p"]), _ptr(e["c3osc"]), b, h, dst,
                                                               block.cout * ps, _ptr(lranges[li]), _ptr(brange), ps,
                                                               ctypes.c_void_p(psc.data_ptr() + 4 * c),
                                                               ctypes.c_void_p(psh.data_ptr() + 4 * c),
                                                               ctypes.c_void_p(pooled.data_ptr() + 4 * c * hw2),
                                                               block.cout * hw2, st),
                       "mirx_conv3x3_direct_terms_nchw_pool")
        else:
            _lib.check(lib.mirx_conv3x3_direct_terms_nchw(_ptr(y), _ptr(e["c3w2p"]), _ptr(e["c3osc"]), b, h, dst,
                                                          block.cout * ps, _ptr(lranges[li]), _ptr(brange), ps, st),
                       "mirx_conv3x3_direct_terms_nchw")
        c += GROWTH
    return buf


def _transition_h2(buf, side, cache, brange, next_buf, next_range, timer=None, pooled=None, pooled_from=0):
    """norm -> relu -> avgpool2 -> conv 1x1 (the pool commutes with the linear conv; two fp16 terms; the pooled values are
    averages of relu(bn(x)), so max|scale| * range + max|shift| bounds them) written into the channel prefix of the next
    block's buffer [B, C', its plane stride], whose range row receives the output ranges: a bn + relu + avgpool pass feeds
    the plain 1x1 conv.  (A one-launch form with the pool inside the conv's staging measured equal in round 2 -- every
    128-channel output tile re-reads the un-pooled map -- and was dropped.)
    `pooled` given: the block's 3x3 launches have already written the channels from `pooled_from` on (their pooled twin);
    the pooling pass then covers only the block's first `pooled_from` channels."""
    lib = _lib.load()
    b, c, ps = buf.shape
    h = w = side
    st = _stream(buf.device)
    hw2 = (h // 2) * (w // 2)
    nps = next_buf.shape[2]
    if pooled is None:
        pooled = torch.empty((b, c, h // 2, w // 2), dtype=torch.float32, device=buf.device)
        _lib.check(lib.mirx_bn_relu_avgpool2(_ptr(buf), c * ps, _ptr(cache["sc"]), _ptr(cache["sh"]), b, c, h, w, _ptr(pooled), ps,
                                             st), "mirx_bn_relu_avgpool2")
    else:
        _lib.check(lib.mirx_bn_relu_avgpool2_into(_ptr(buf), c * ps, _ptr(cache["sc"]), _ptr(cache["sh"]), b, pooled_from, h, w,
                                                  _ptr(pooled), c * hw2, ps, st), "mirx_bn_relu_avgpool2_into")
    ev = _timer_start(timer)
    _lib.check(lib.mirx_conv1x1_bn_relu_split2h(_ptr(pooled), c * hw2, c, None, None, _ptr(cache["w2"]), _ptr(cache["osc"]),
                                                None, b, hw2, c // 2, 0, _ptr(next_buf), next_buf.shape[1] * nps,
                                                _ptr(brange), cache["ks"], cache["kb"], _ptr(next_range), 0, nps, st),
               "mirx_conv1x1_bn_relu_split2h")
    _timer_stop(timer, ev, 2.0 * b * hw2 * c * (c // 2), 4.0 * b * hw2 * (c + c // 2), "conv1x1")
    return next_buf


class _Transition(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm = nn.BatchNorm2d(cin)
        self.relu = nn.ReLU(inplace=True)
        self.conv = nn.Conv2d(cin, cout, kernel_size=1, stride=1, bias=False)
        self.pool = nn.AvgPool2d(kernel_size=2, stride=2)


def _make_features():
    layers = OrderedDict()
    layers["conv0"] = nn.Conv2d(3, INIT_FEATURES, kernel_size=7, stride=2, padding=3, bias=False)
    layers["norm0"] = nn.BatchNorm2d(INIT_FEATURES)
    layers["relu0"] = nn.ReLU(inplace=True)
    layers["pool0"] = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
    c = INIT_FEATURES
    for i, n in enumerate(BLOCK_CONFIG):
        blk = _DenseBlock(n, c)
        layers[f"denseblock{i + 1}"] = blk
        c = blk.cout
        if i != len(BLOCK_CONFIG) - 1:
            layers[f"transition{i + 1}"] = _Transition(c, c // 2)
            c //= 2
    layers["norm5"] = nn.BatchNorm2d(c)
    feats = nn.Sequential(layers)
    for m in feats.modules():          # the published initialisation
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
    return feats, c


def _winograd_weights(w):
    """conv2 weights [32, 128, 3, 3] -> U = G g G^T of Winograd F(2x2,3x3), laid out for
    mirx_conv3x3_winograd_nchw: [stage = c // 8][xi = 4 i + j][c % 8][oc], fp32."""
    g = w.detach().float()
    gm = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], device=g.device)
    u = torch.einsum("ik,ockl,jl->ocij", gm, g, gm)                       # [oc, c, 4, 4]
    oc, cin = u.shape[0], u.shape[1]
    u = u.reshape(oc, cin // 8, 8, 16).permute(1, 3, 2, 0)                # [stage, xi, c % 8, oc]
    return u.contiguous()


def _winograd_weights_split3(w):
    """conv2 weights [32, 128, 3, 3] -> the three bf16 terms of U = G g G^T, laid out for
    mirx_conv3x3_winograd_split3_nchw: [stage = c // 16][xi = 4 i + j][term][oc][c % 16] bf16."""
    g = w.detach().float()
    gm = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], device=g.device)
    u = torch.einsum("ik,ockl,jl->ocij", gm, g, gm)                       # [oc, c, 4, 4]
    h = u.to(torch.bfloat16)
    m = (u - h.float()).to(torch.bfloat16)
    lo = (u - h.float() - m.float()).to(torch.bfloat16)
    oc, cin = u.shape[0], u.shape[1]
    t = torch.stack([h, m, lo], 0).reshape(3, oc, cin // 16, 16, 16)      # [term, oc, stage, c % 16, xi]
    return t.permute(2, 4, 0, 1, 3).contiguous()


def _conv3x3_weights_split3(w):
    """conv2 weights [32, 128, 3, 3] -> the three bf16 terms laid out for mirx_conv3x3_direct_split3_nchw:
    [stage = c // 16][tap = 3 ky + kx][term][oc][c % 16] bf16."""
    w = w.detach().float()
    h = w.to(torch.bfloat16)
    m = (w - h.float()).to(torch.bfloat16)
    lo = (w - h.float() - m.float()).to(torch.bfloat16)
    oc, cin = w.shape[0], w.shape[1]
    t = torch.stack([h, m, lo], 0).reshape(3, oc, cin // 16, 16, 9)        # [term, oc, stage, c % 16, tap]
    return t.permute(2, 4, 0, 1, 3).contiguous()


# Channel order of the pre-split bottleneck (mirx_conv1x1_bn_relu_split2h_terms): slot j of group g holds channel
# 64 (g >> 2) + 32 ((g >> 1) & 1) + 4 (g & 1) + (j & 3) + 8 (j >> 2) -- the 16 channels one lane of the 1x1-conv kernel owns.
YTERMS_CHANNEL_ORDER = [64 * (g >> 2) + 32 * ((g >> 1) & 1) + 4 * (g & 1) + (j & 3) + 8 * (j >> 2)
                        for g in range(8) for j in range(16)]


def _conv3x3_weights_split2h(w, channel_order=None):
    """conv2 weights [32, 128, 3, 3] -> (w2, oscale): the two fp16 terms of W[oc] * ws[oc] (ws a power of two per output
    channel) laid out for mirx_conv3x3_direct_split2h_nchw: [stage = c // 16][tap = 3 ky + kx][term][oc][c % 16] fp16, and
    oscale = 1 / ws fp32 [32].  channel_order: input channels re-ordered first (YTERMS_CHANNEL_ORDER for the terms path)."""
    w = w.detach().float()
    if channel_order is not None:
        w = w[:, torch.as_tensor(channel_order, device=w.device)]
    oc, cin = w.shape[0], w.shape[1]
    ws = _pow2_row_scales(w.reshape(oc, -1))
    wf = w * ws.view(-1, 1, 1, 1)
    h = wf.to(torch.float16)
    lo = (wf - h.float()).to(torch.float16)
    t = torch.stack([h, lo], 0).reshape(2, oc, cin // 16, 16, 9)           # [term, oc, stage, c % 16, tap]
    return t.permute(2, 4, 0, 1, 3).contiguous(), (1.0 / ws).contiguous()


def _stem_weights_split3(w):
    """conv0 weights [64, 3, 7, 7] -> the three bf16 terms laid out for mirx_stem_conv7_bn_relu_pool_split3:
    [2 oc blocks][11 steps][term][32 oc][16 k]; k = 8 g + i of step s is row (c, ky) = divmod(2 s + g, 7) and
    kx = 0, 2, 4, 6, 1, 3, 5, (7 = zero) for i = 0..7; row 21 is zero."""
    w = w.detach().float()
    wk = torch.zeros((64, 22, 8), dtype=torch.float32, device=w.device)
    rows = w.reshape(64, 21, 7)                                            # [oc, (c, ky), kx]
    wk[:, :21, 0:4] = rows[:, :, 0::2]
    wk[:, :21, 4:7] = rows[:, :, 1::2]
    wk = wk.reshape(64, 11, 16)
    h = wk.to(torch.bfloat16)
    m = (wk - h.float()).to(torch.bfloat16)
    lo = (wk - h.float() - m.float()).to(torch.bfloat16)
    t = torch.stack([h, m, lo], 0).reshape(3, 2, 32, 11, 16)               # [term, block, oc, step, k]
    return t.permute(1, 3, 0, 2, 4).contiguous()


def _stem_weights_split2h(w):
    """conv0 weights [64, 3, 7, 7] -> (w2, oscale): the two fp16 terms of W[oc] * ws[oc] (a power of two per output channel)
    in the K order of mirx_stem_conv7_bn_relu_pool_split3 -- [2 oc blocks][11 steps][term][32 oc][16 k] -- and 1 / ws [64]."""
    w = w.detach().float()
    ws = _pow2_row_scales(w.reshape(64, -1))
    wk = torch.zeros((64, 22, 8), dtype=torch.float32, device=w.device)
    rows = (w * ws.view(-1, 1, 1, 1)).reshape(64, 21, 7)                   # [oc, (c, ky), kx]
    wk[:, :21, 0:4] = rows[:, :, 0::2]
    wk[:, :21, 4:7] = rows[:, :, 1::2]
    wk = wk.reshape(64, 11, 16)
    h = wk.to(torch.float16)
    lo = (wk - h.float()).to(torch.float16)
    t = torch.stack([h, lo], 0).reshape(2, 2, 32, 11, 16)                  # [term, block, oc, step, k]
    return t.permute(1, 3, 0, 2, 4).contiguous(), (1.0 / ws).contiguous()


def _split3_weights(w):
    """[cout, cin] fp32 -> the three bf16 terms of every weight (w = h + m + l exactly, 3 x 8 mantissa
    bits), laid out for mirx_conv1x1_bn_relu_split3: [cout // 128][cin // 16][3][128][16] bf16."""
    w = w.detach().float()
    if w.shape[0] % 128:                               # Linear layers: zero rows up to the next output tile
        w = F.pad(w, (0, 0, 0, 128 - w.shape[0] % 128))
    h = w.to(torch.bfloat16)
    m = (w - h.float()).to(torch.bfloat16)
    lo = (w - h.float() - m.float()).to(torch.bfloat16)
    cout, cin = w.shape
    t = torch.stack([h, m, lo], 0).reshape(3, cout // 128, 128, cin // 16, 16)
    return t.permute(1, 3, 0, 2, 4).contiguous()


def _pow2_row_scales(w):
    """Per output row of w [cout, k]: the power of two that puts the row's largest |w| in [2^13, 2^14) (so that the low
    fp16 term of all but vanishing weights stays a normal number); rows of zeros (or non-finite rows) get 1."""
    amax = w.abs().amax(dim=1)
    ok = torch.isfinite(amax) & (amax > 0)
    e = torch.floor(torch.log2(torch.where(ok, amax, torch.ones_like(amax))))
    # floor(log2) of an fp32 can be off by one at exact powers of two after rounding: fix up
    e = torch.where(torch.exp2(e) > amax, e - 1, e)
    e = torch.where(torch.exp2(e + 1) <= amax, e + 1, e)
    return torch.where(ok, torch.exp2(13 - e), torch.ones_like(amax))


def _split2h_weights(w):
    """[cout, cin] fp32 -> (w2, oscale): the two fp16 terms of W[o, :] * ws[o] (ws a power of two per output row),
    laid out for mirx_conv1x1_bn_relu_split2h: [cout // 128][cin // 16][2][128][16] fp16, and oscale = 1 / ws fp32."""
    w = w.detach().float()
    ws = _pow2_row_scales(w)
    wf = w * ws[:, None]
    h = wf.to(torch.float16)
    lo = (wf - h.float()).to(torch.float16)
    cout, cin = w.shape
    t = torch.stack([h, lo], 0).reshape(2, cout // 128, 128, cin // 16, 16)
    return t.permute(1, 3, 0, 2, 4).contiguous(), (1.0 / ws).contiguous()


def _bn_affine(bn):
    """Eval-mode BatchNorm as y = x*scale + shift (fp32)."""
    scale = bn.weight.detach().float() * torch.rsqrt(bn.running_var.detach().float() + bn.eps)
    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    return scale.contiguous(), shift.contiguous()


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _linear_s3_ok(mod, x):
    return (_cfg(mod).linear_three_bf16 and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()
            and mod.in_features % 16 == 0)


def _linear_w3(mod):
    """The three-term bf16 image of a Linear's weight, rebuilt when the parameter changes."""
    w = mod.weight
    key = (w.data_ptr(), w._version, w.device)
    cached = getattr(mod, "_mirx_w3", None)
    if cached is None or cached[0] != key:
        cached = (key, _split3_weights(w.detach()))
        mod._mirx_w3 = cached
    return cached[1]


def _linear_h2_weights(mod):
    """(w2, w_scale): the two fp16 terms of W * w_scale, w_scale the power of two that puts the largest |w| in
    [2^13, 2^14) (so the low term stays a normal fp16 number), laid out [ceil(n / 128)][k / 16][2][128][16]."""
    w = mod.weight
    key = (w.data_ptr(), w._version, w.device)
    cached = getattr(mod, "_mirx_w2", None)
    if cached is None or cached[0] != key:
        wf = w.detach().float()
        amax = float(wf.abs().max())
        ws = 2.0 ** math.floor(math.log2(16384.0 / amax)) if amax > 0 and math.isfinite(amax) else 1.0
        wf = wf * ws
        if wf.shape[0] % 128:
            wf = F.pad(wf, (0, 0, 0, 128 - wf.shape[0] % 128))
        wh = wf.to(torch.float16)
        wl = (wf - wh.float()).to(torch.float16)
        n, k = wf.shape
        t = torch.stack([wh, wl], 0).reshape(2, n // 128, 128, k // 16, 16)
        cached = (key, t.permute(1, 3, 0, 2, 4).contiguous(), ws)
        mod._mirx_w2 = cached
    return cached[1], cached[2]


def _terms_of(a, scale):
    """Host-side "terms rows" of `a` [rows, k] * scale (include/mirx.h, mirx_linear_terms): fp16 [rows, ceil32(k) // 32, 2, 32]
    = per 32-feature group the high terms, then the low terms.  Used for weights (once per layer); activations are converted
    on the device (mirx_rows_to_terms, mirx_layernorm_terms, the terms output of mirx_linear_terms)."""
    a = a.float() * scale
    rows, k = a.shape
    if k % 32:
        a = F.pad(a, (0, 32 - k % 32))
    hi = a.to(torch.float16)
    lo = (a - hi.float()).to(torch.float16)
    return torch.stack([hi.view(rows, -1, 32), lo.view(rows, -1, 32)], 2).contiguous()


def _linear_terms_weights(mod):
    """(wt, w_scale): terms rows of W * w_scale, rows padded to a multiple of 256 with zeros; w_scale the power of two that
    puts the largest |w| in [2^13, 2^14) (the low term stays a normal fp16 number)."""
    w = mod.weight
    key = _tkey(w)
    cached = getattr(mod, "_mirx_wt", None)
    if cached is None or cached[0] != key:
        wf = w.detach().float()
        amax = float(wf.abs().max())
        ws = 2.0 ** math.floor(math.log2(16384.0 / amax)) if amax > 0 and math.isfinite(amax) else 1.0
        if wf.shape[0] % 256:
            wf = F.pad(wf, (0, 0, 0, 256 - wf.shape[0] % 256))
        cached = (key, _terms_of(wf, ws), ws)
        mod._mirx_wt = cached
    return cached[1], cached[2]


def _terms_scale(bound):
    """The power of two that brings |x| <= bound to at most 2^15 (fp16 terms overflow at 65504)."""
    return 2.0 ** math.floor(math.log2(32768.0 / bound))


def _rows_to_terms(x, bound):
    """[HIP] terms rows of x [..., k] (fp32, last axis contiguous) scaled for |x| <= bound -> (xt fp16 [m, ceil32(k) * 2], scale)."""
    k = x.shape[-1]
    x = x.reshape(-1, k)
    if x.stride(-1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16:
        x = x.contiguous()
    m = x.shape[0]
    xs = _terms_scale(bound)
    xt = torch.empty(m, (k + 31) // 32 * 64, dtype=torch.float16, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().mirx_rows_to_terms(_ptr(x), m, k, x.stride(0), xs, _ptr(xt), _stream(x.device)), "mirx_rows_to_terms")
    return xt, xs


def _layernorm_terms(ln, x, bound):
    """[HIP] LayerNorm over the last axis written as terms rows (the next Linear's input) -> (xt, scale)."""
    c = x.shape[-1]
    x = x.contiguous()
    m = x.numel() // c
    xs = _terms_scale(bound)
    xt = torch.empty(m, (c + 31) // 32 * 64, dtype=torch.float16, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().mirx_layernorm_terms(_ptr(x), m, c, _ptr(ln.weight.detach()) if ln.weight is not None else None,
                                                    _ptr(ln.bias.detach()) if ln.bias is not None else None, float(ln.eps), xs,
                                                    _ptr(xt), _stream(x.device)), "mirx_layernorm_terms")
    return xt, xs


def _linear_terms(mod, xt, xs, lead, act=0, res=None, gamma=None, out=None, terms_bound=None):
    """[HIP] y = epi(x W^T + b) through mirx_linear_terms; xt / xs = the input as terms rows and its scale, `lead` the
    leading shape of the result.  terms_bound: the result is returned as (terms rows, scale) for |y| <= terms_bound instead
    of fp32 (the fc1 -> fc2 hand-over)."""
    wt, ws = _linear_terms_weights(mod)
    m = xt.shape[0]
    n = mod.out_features
    dev = xt.device
    yt = ys = None
    if terms_bound is not None:
        ys = _terms_scale(terms_bound)
        yt = torch.empty(m, (n + 31) // 32 * 64, dtype=torch.float16, device=dev)
    elif out is None:
        out = torch.empty(tuple(lead) + (n,), dtype=torch.float32, device=dev)
    if res is not None:
        assert res.is_contiguous() and res.shape == out.shape
    lib = _lib.load()
    wsb = int(lib.mirx_linear_terms_workspace_bytes(m, mod.in_features, n)) if _cfg(mod).linear_terms_split_tail else 0
    wsp = torch.empty(wsb, dtype=torch.uint8, device=dev) if wsb else None         # stream-ordered: safe across streams
    with torch.cuda.device(dev):
        _lib.check(_lib.load().mirx_linear_terms(_ptr(xt), m, mod.in_features, _ptr(wt),
                                                 _ptr(mod.bias.detach()) if mod.bias is not None else None, n, act,
                                                 _ptr(res) if res is not None else None,
                                                 _ptr(gamma.detach()) if gamma is not None else None, 1.0 / (xs * ws),
                                                 _ptr(out) if yt is None else None, _ptr(yt) if yt is not None else None,
                                                 ys if ys is not None else 1.0, _ptr(wsp) if wsp is not None else None, wsb,
                                                 _stream(dev)), "mirx_linear_terms")
    return (yt, ys) if yt is not None else out


def _linear_terms_ok(mod, rows, lins, bounds):
    """The DMA-fed Linear (mirx_linear_terms) serves a block when the batch fills its 256 x 256 tiles and every input has a
    provable bound (the same contract as _linear_h2_ok)."""
    cfg = _cfg(mod)
    return (cfg.linear_two_fp16 and cfg.linear_three_bf16 and cfg.linear_terms_min_rows > 0 and rows >= cfg.linear_terms_min_rows
            and all(l.out_features % 4 == 0 for l in lins) and all(math.isfinite(b) and 0.0 < b < 3.0e4 for b in bounds))


def _tkey(t):
    """Cache key of a parameter: a replaced tensor object (load_state_dict(assign=True), `m.weight = nn.Parameter(..)`,
    `p.data = ..`) starts again at version 0, so the version alone would keep a stale entry alive."""
    return None if t is None else (id(t), t.data_ptr(), t._version)


def _layernorm_bound(ln):
    """max |LayerNorm(x)_j| over all inputs: |x_j - mean| / std <= sqrt(C - 1), so sqrt(C - 1) max|gamma| + max|beta|."""
    g, b = ln.weight, ln.bias
    key = (_tkey(g), _tkey(b))
    cached = getattr(ln, "_mirx_bound", None)
    if cached is None or cached[0] != key:
        c = ln.normalized_shape[-1]
        gm = 1.0 if g is None else float(g.detach().abs().max())
        bm = 0.0 if b is None else float(b.detach().abs().max())
        cached = (key, math.sqrt(max(c - 1, 1)) * gm + bm)
        ln._mirx_bound = cached
    return cached[1]


def _linear_out_bound(ln, lin, rows=None):
    """max |lin(LayerNorm(x))_j| over all inputs, for output rows `rows` (slice) of `lin`:
    ||LN(x)||_2 <= max|gamma| sqrt(C) + ||beta||_2 (the normalised vector has squared norm C), so
    |y_j| <= that * ||W_j||_2 + |b_j|.  GELU and softmax-weighted averages of such outputs obey the same bound."""
    g, b = ln.weight, ln.bias
    ver = (_tkey(g), _tkey(b), _tkey(lin.weight), _tkey(lin.bias))
    store = lin.__dict__.setdefault("_mirx_out_bound", {})
    if store.get("ver") != ver:                      # weights changed: every cached row range is stale
        store.clear()
        store["ver"] = ver
    key = None if rows is None else (rows.start, rows.stop)
    if key not in store:
        c = ln.normalized_shape[-1]
        gm = 1.0 if g is None else float(g.detach().abs().max())
        bn = 0.0 if b is None else float(torch.linalg.vector_norm(b.detach().double()))
        w = lin.weight.detach() if rows is None else lin.weight.detach()[rows]
        wn = float(torch.linalg.vector_norm(w.double(), dim=1).max())
        bb = 0.0 if lin.bias is None else float((lin.bias.detach() if rows is None else lin.bias.detach()[rows]).abs().max())
        store[key] = (gm * math.sqrt(c) + bn) * wn + bb
    return store[key]


def _linear_h2_ok(mod, x, bound):
    return (_cfg(mod).linear_two_fp16 and _linear_s3_ok(mod, x) and math.isfinite(bound) and 0.0 < bound < 3.0e4)


def _linear_h2(mod, x, bound, act=0, res=None, gamma=None, out=None):
    """[HIP] y = epi(x W^T + b) through mirx_linear_split2h; `bound` >= max |x| (the caller's proof obligation: fp16
    terms overflow at 65504) -- x is scaled by the power of two that brings `bound` to at most 2^15."""
    w2, ws = _linear_h2_weights(mod)
    xs = 2.0 ** math.floor(math.log2(32768.0 / bound))
    x = x.contiguous()
    m = x.numel() // mod.in_features
    if out is None:
        out = torch.empty(x.shape[:-1] + (mod.out_features,), dtype=torch.float32, device=x.device)
    if res is not None:
        assert res.is_contiguous() and res.shape == out.shape
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().mirx_linear_split2h(_ptr(x), m, mod.in_features, _ptr(w2),
                                                   _ptr(mod.bias.detach()) if mod.bias is not None else None,
                                                   mod.out_features, act, _ptr(res) if res is not None else None,
                                                   _ptr(gamma.detach()) if gamma is not None else None,
                                                   xs, 1.0 / (xs * ws), _ptr(out), _stream(x.device)),
                   "mirx_linear_split2h")
    return out


def _linear_s3(mod, x, act=0, res=None, gamma=None, out=None):
    """[HIP] y = epi(x W^T + b) through mirx_linear_split3 (include/mirx.h); x: [..., k] fp32 CUDA.
    act=1: GELU; res/gamma: y = res + gamma * v (may be written in place with out=res)."""
    w3 = _linear_w3(mod)
    x = x.contiguous()
    m = x.numel() // mod.in_features
    if out is None:
        out = torch.empty(x.shape[:-1] + (mod.out_features,), dtype=torch.float32, device=x.device)
    if res is not None:
        assert res.is_contiguous() and res.shape == out.shape
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().mirx_linear_split3(_ptr(x), m, mod.in_features, _ptr(w3),
                                                  _ptr(mod.bias.detach()) if mod.bias is not None else None,
                                                  mod.out_features, act, _ptr(res) if res is not None else None,
                                                  _ptr(gamma.detach()) if gamma is not None else None, _ptr(out),
                                                  _stream(x.device)), "mirx_linear_split3")
    return out


def _linear_auto(mod, x, act=0):
    """nn.Linear forward that takes the MFMA kernel for CUDA fp32 inference (any [..., k] input, k % 16 == 0) and the
    module's own forward otherwise.  act: 0 none, 1 erf-GELU, 2 tanh-GELU (fused in the kernel's epilogue)."""
    if _linear_s3_ok(mod, x):
        return _linear_s3(mod, x, act=act)
    y = mod(x)
    return y if act == 0 else F.gelu(y, approximate="tanh" if act == 2 else "none")


def _layernorm(ln, x, tokens_per_image=0):
    """nn.LayerNorm over the last axis: [HIP] mirx_layernorm for CUDA fp32 inference, F.layer_norm otherwise.
    tokens_per_image > 0: x is [images * tpi, c] and the result comes back channels-first [images, c, tpi]."""
    c = ln.normalized_shape[-1]
    if (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and c % 4 == 0 and c <= 8192
            and len(ln.normalized_shape) == 1 and x.shape[-1] == c):          # (the kernel keeps a row in registers: c <= 8192)
        x = x.contiguous()
        m = x.numel() // c
        if tokens_per_image:
            out = torch.empty((m // tokens_per_image, c, tokens_per_image), dtype=torch.float32, device=x.device)
        else:
            out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().mirx_layernorm(_ptr(x), m, c, _ptr(ln.weight.detach()) if ln.weight is not None else None,
                                                  _ptr(ln.bias.detach()) if ln.bias is not None else None, float(ln.eps),
                                                  _ptr(out), int(tokens_per_image), _stream(x.device)), "mirx_layernorm")
        return out
    y = F.layer_norm(x, ln.normalized_shape, ln.weight, ln.bias, ln.eps)
    if tokens_per_image:
        y = y.view(-1, tokens_per_image, c).transpose(1, 2).contiguous()
    return y


class _ConvAsLinear:
    """A Conv2d whose kernel equals its stride (non-overlapping patches) seen as a Linear over patch rows: weight =
    conv.weight.flatten(1) zero-padded to a multiple of 16 input features (duck-typed for _linear_s3 / _linear_h2)."""

    def __init__(self, conv, channels_last=False):
        self.conv = conv
        self.channels_last = channels_last            # feature order (ky, kx, c): patch rows gathered from an NHWC map
        self._key = None

    def refresh(self):
        w, b = self.conv.weight, self.conv.bias
        key = (w.data_ptr(), w._version, None if b is None else b._version)
        if key != self._key:
            flat = (w.detach().permute(0, 2, 3, 1) if self.channels_last else w.detach()).flatten(1)
            k = flat.shape[1]
            self.k = k
            self.in_features = (k + 15) // 16 * 16
            self.out_features = flat.shape[0]
            self.weight = F.pad(flat, (0, self.in_features - k)) if self.in_features != k else flat.contiguous()
            self.bias = None if b is None else b.detach()
            self._key = key
        return self


def _conv_patch_tokens(conv, x, ln2d=None, nchw_out=False):
    """[HIP] Conv2d(kernel = stride = p, no padding) on NCHW x as mirx_patchify_nchw + the MFMA Linear kernel.
    ln2d: an nn.LayerNorm applied over the channel axis of every pixel first (timm LayerNorm2d in front of the ConvNeXt
    downsample conv) -- its output is bounded, so the Linear runs on two fp16 terms.  -> [B * gh * gw, cout] rows, or the
    NCHW map [B, cout, gh, gw] when nchw_out."""
    pl = conv.__dict__.get("_mirx_as_linear")
    if pl is None:
        pl = _ConvAsLinear(conv)
        conv.__dict__["_mirx_as_linear"] = pl
    pl.__dict__["_mirx_cfg"] = _cfg(conv)          # the view runs under its conv's configuration
    pl.refresh()
    p = conv.kernel_size[0]
    b, c, h, w = x.shape
    gh, gw = h // p, w // p
    x = x.contiguous()
    rows = torch.empty((b * gh * gw, pl.in_features), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        st = _stream(x.device)
        _lib.check(lib.mirx_patchify_nchw(_ptr(x), b, c, h, w, p, _ptr(ln2d.weight.detach()) if ln2d is not None else None,
                                          _ptr(ln2d.bias.detach()) if ln2d is not None else None,
                                          float(ln2d.eps) if ln2d is not None else 0.0, _ptr(rows), pl.in_features, st),
                   "mirx_patchify_nchw")
        if nchw_out:
            out = torch.empty((b, pl.out_features, gh, gw), dtype=torch.float32, device=x.device)
            bound = _layernorm_bound(ln2d) if ln2d is not None else float("inf")
            if _linear_h2_ok(pl, rows, bound):                 # the rows are LayerNorm outputs: two fp16 terms
                w2, ws = _linear_h2_weights(pl)
                _lib.check(lib.mirx_linear_split2h_nchw(_ptr(rows), b, gh * gw, pl.in_features, _ptr(w2),
                                                        _ptr(pl.bias) if pl.bias is not None else None, pl.out_features, None,
                                                        None, float(bound), None, 1.0 / ws, _ptr(out), st),
                           "mirx_linear_split2h_nchw")
                return out
            _lib.check(lib.mirx_linear_split3_nchw(_ptr(rows), b, gh * gw, pl.in_features, _ptr(_linear_w3(pl)),
                                                   _ptr(pl.bias) if pl.bias is not None else None, pl.out_features, None, None,
                                                   _ptr(out), st), "mirx_linear_split3_nchw")
            return out
    if ln2d is not None:
        bound = _layernorm_bound(ln2d)
        if _linear_h2_ok(pl, rows, bound):
            return _linear_h2(pl, rows, bound)
    return _linear_s3(pl, rows)


_TENSOR_VERSION = operator.attrgetter("_version")


class DenseNet121(_Configurable, nn.Module):
    """Reference model.py:42-84, MI355X-native inference path."""
    accepts_uint8 = True        # forward() takes raw 8-bit images and applies ToTensor + Normalize itself (input_mean / input_std)

    def __init__(self, pretrained=False, embedding_dim=None, num_labels=None, weights=None):
        super().__init__()
        if pretrained and weights is None:
            raise RuntimeError("pretrained=True needs a download in the reference (model.py:53); "
                               "pass weights=<state dict or path> instead")
        feats, in_features = _make_features()
        self.densenet121 = nn.Sequential(feats)
        # reference model.py:59-60: ReLU appended to the feature stack, avgpool to the wrapper
        self.densenet121[0].add_module("relu", nn.ReLU(inplace=True))
        self.densenet121.add_module("avgpool", nn.AdaptiveAvgPool2d((1, 1)))
        self.fc = nn.Linear(in_features, embedding_dim) if embedding_dim else None
        out_features = embedding_dim if embedding_dim else in_features
        self.classification_head = nn.Linear(out_features, num_labels) if num_labels else None
        self.conv1x1_timer = None          # list -> (start event, stop event, FLOP) per fused conv launch
        # raw 8-bit input: forward() also takes uint8 [B, 3, H, W] and applies the reference's ToTensor + Normalize
        # (test.py:1309-1332) with these constants -- inside the stem kernel at 224 x 224, a quarter of the PCIe / HBM bytes
        self.register_buffer("input_mean", torch.tensor(IMAGENET_MEAN, dtype=torch.float32), persistent=False)
        self.register_buffer("input_std", torch.tensor(IMAGENET_STD, dtype=torch.float32), persistent=False)
        self._infer_cache = None           # folded BatchNorm parameters of the inference path
        self._init_config()
        if weights is not None:
            sd = torch.load(weights, map_location="cpu") if isinstance(weights, str) else weights
            for key in ("state-dict", "state_dict"):      # wrappers test.py:1273-1276 accepts
                if isinstance(sd, dict) and key in sd:
                    sd = sd[key]
            self.load_state_dict(sd, strict=False)

    # -- the plain module graph (training, CPU tensors): same ops as the reference -------------
    def forward_eager(self, x):
        x = self.densenet121(x)
        return torch.flatten(x, 1)

    # -- MI355X inference path -------------------------------------------------------------------
    # The folded / pre-split weights are derived from the parameters and BatchNorm buffers; they are rebuilt whenever
    # any of those changed.  Routes that announce themselves drop the cache at once: load_state_dict (this module's own
    # and, through the pre-hook, a parent's), _apply (device moves, dtype casts), train().  What remains -- an in-place
    # edit of a parameter, a replaced tensor object -- is caught by ONE pass per forward over the watched tensors
    # (versions and identities: ~0.1 ms for the 604 tensors; it was two passes of 0.22 ms, 18 % of a one-image forward).
    def _watch_lists(self):
        w = self.__dict__.get("_mirx_watch")
        if w is None:
            dicts, keys = [], []
            for m in self.densenet121[0].modules():
                for d in (m._parameters, m._buffers):
                    for k in d:
                        if d[k] is not None and k != "num_batches_tracked":       # not read by an eval-mode BatchNorm
                            dicts.append(d)
                            keys.append(k)
            w = (dicts, keys, [d[k] for d, k in zip(dicts, keys)])
            self.__dict__["_mirx_watch"] = w
        return w

    def _weights_version(self):
        """(sum of tensor versions, sum of tensor identities) over every parameter and BatchNorm buffer of the feature
        stack: in-place edits bump the first, a REPLACED tensor object (`conv.weight = nn.Parameter(..)`) changes the second."""
        dicts, keys, tens = self._watch_lists()
        ident = sum(map(id, map(dict.__getitem__, dicts, keys)))
        if ident != self.__dict__.get("_mirx_watch_ident"):
            self.__dict__.pop("_mirx_watch", None)                 # a tensor object was replaced: take the new objects
            dicts, keys, tens = self._watch_lists()
            self.__dict__["_mirx_watch_ident"] = ident = sum(map(id, tens))
        return sum(map(_TENSOR_VERSION, tens)), ident

    def _cache(self):
        ver = self._weights_version()
        cache = self._infer_cache
        if cache is None or cache.get("_version") != ver:
            cache = self._prepare_inference()
            cache["_version"] = ver
        if cache.get("_cuda"):
            # the folded / split weights are built by kernels on the stream that was current at the time; a stream that
            # has not seen them yet (bench.py's embed streams) waits for the newest build's event -- no device-wide sync
            sid = torch.cuda.current_stream().stream_id
            if sid not in cache["_seen"]:
                torch.cuda.current_stream().wait_event(cache["_built"])
                cache["_seen"].add(sid)
        return cache

    @staticmethod
    def _mark_built(cache):
        """Call after enqueueing work that fills `cache` on the current stream."""
        if cache.get("_cuda"):
            ev = torch.cuda.Event()
            ev.record()
            cache["_built"] = ev
            cache["_seen"] = {torch.cuda.current_stream().stream_id}

    def _drop_cache(self, *a, **k):
        self._infer_cache = None
        self.__dict__.pop("_mirx_watch", None)

    def train(self, mode=True):
        self._drop_cache()
        return super().train(mode)

    def load_state_dict(self, *a, **k):
        self._drop_cache()
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._drop_cache()
        return super()._apply(fn, *a, **k)

    def _prepare_inference(self):
        """Fold every eval-mode BatchNorm once: norm1/transition norm -> (scale, shift) for the
        fused HIP passes; norm2 -> into conv1's weights and bias.  The operands of the legacy (three-bf16-term /
        Winograd) kernels are NOT built here: _ensure_legacy() does that when an input takes that path."""
        f = self.densenet121[0]
        cache = {}
        for name, m in f.named_children():
            if name.startswith("denseblock"):
                blk = {}
                for lname, layer in m.items():
                    sc1, sh1 = _bn_affine(layer.norm1)
                    sc2, sh2 = _bn_affine(layer.norm2)
                    w1 = (layer.conv1.weight.detach().float() * sc2.view(-1, 1, 1, 1)).contiguous()
                    w1t = w1.view(w1.shape[0], w1.shape[1]).t().contiguous()       # [cin, 128] for the HIP GEMM
                    blk[lname] = (sc1, sh1, w1, sh2, torch.ones_like(sh2), w1t, None, None)
                cache[name] = blk
            elif name.startswith("transition"):
                wt = m.conv.weight.detach().float()
                cache[name] = _bn_affine(m.norm) + (wt.view(wt.shape[0], wt.shape[1]).t().contiguous(), None)
        cache["norm0"] = _bn_affine(f.norm0)
        cache["norm5"] = _bn_affine(f.norm5)
        cache["_cuda"] = f.conv0.weight.is_cuda
        self._infer_cache = cache
        self._mark_built(cache)
        return cache

    def _ensure_legacy(self, cache):
        """Operands of the kernels that serve inputs other than 224 x 224 (three bf16 terms, Winograd): built on first use."""
        if cache.get("_legacy"):
            return
        f = self.densenet121[0]
        for name, m in f.named_children():
            if name.startswith("denseblock"):
                for lname, layer in m.items():
                    e = cache[name][lname]
                    w1 = e[2]
                    cache[name][lname] = e[:6] + ((_winograd_weights(layer.conv2.weight), _winograd_weights_split3(layer.conv2.weight),
                                                   _conv3x3_weights_split3(layer.conv2.weight)),
                                                  _split3_weights(w1.view(w1.shape[0], w1.shape[1])))
            elif name.startswith("transition"):
                wt = m.conv.weight.detach().float()
                cache[name] = cache[name][:3] + (_split3_weights(wt.view(wt.shape[0], wt.shape[1])),)
        cache["_legacy"] = True
        self._mark_built(cache)

    def _features_fused(self, x, cache):
        """-> (feature map before norm5 [B,1024,h,w])"""
        f = self.densenet121[0]
        lib = _lib.load()
        if x.dtype == torch.uint8:
            if self._h2_ok(x):
                return self._features_h2(x.contiguous(), cache)          # normalised inside the stem kernel
            x = self.normalize_uint8(x)
        x = x.contiguous().float()
        b, _, h, w = x.shape
        if self._h2_ok(x):
            return self._features_h2(x, cache)
        self._ensure_legacy(cache)
        cfg = _cfg(self)
        if cfg.hip_stem and h % 4 == 0 and w % 4 == 0 and h >= 8 and w >= 8:
            sc, sh = cache["norm0"]
            y = torch.empty((b, INIT_FEATURES, h // 4, w // 4), dtype=torch.float32, device=x.device)
            if cfg.stem_three_bf16 and b <= 65535:
                if "conv0_w3" not in cache:
                    cache["conv0_w3"] = _stem_weights_split3(f.conv0.weight)
                    self._mark_built(cache)
                _lib.check(lib.mirx_stem_conv7_bn_relu_pool_split3(_ptr(x), _ptr(cache["conv0_w3"]), _ptr(sc), _ptr(sh),
                                                                   b, h, w, _ptr(y), _stream(x.device)), "mirx_stem_split3")
            else:
                wt = f.conv0.weight.detach().float().contiguous()
                _lib.check(lib.mirx_stem_conv7_bn_relu_pool(_ptr(x), _ptr(wt), _ptr(sc), _ptr(sh), b, h, w,
                                                            _ptr(y), _stream(x.device)), "mirx_stem")
            x = y
        else:
            x = f.pool0(f.relu0(f.norm0(f.conv0(x))))
        children = list(f.named_children())
        for i, (name, m) in enumerate(children):
            if name.startswith("denseblock"):
                x = _dense_block_fused(m, x, cache[name], cfg, self.conv1x1_timer)
            elif name.startswith("transition"):
                nxt = children[i + 1][1] if i + 1 < len(children) and children[i + 1][0].startswith("denseblock") else None
                x = _transition_fused(m, x, cache[name], cfg, self.conv1x1_timer, nxt.cout if nxt is not None else None)
        return x

    def normalize_uint8(self, x):
        """ToTensor + Normalize of the reference (test.py:1309-1332) on a uint8 [B, 3, H, W] batch, the same fp32 operations in
        the same order (u / 255, - mean, / std): what the stem kernel's table holds."""
        m = self.input_mean.to(x.device).view(1, 3, 1, 1)
        s_ = self.input_std.to(x.device).view(1, 3, 1, 1)
        return (x.float() / 255.0 - m) / s_

    def _h2_ok(self, x):
        """The two-fp16-term path covers the geometry of the reference's 224 x 224 evaluation (maps 56 / 28 / 14 / 7)."""
        cfg = _cfg(self)
        return (cfg.densenet_two_fp16 and cfg.hip_stem and cfg.hip_conv1x1 and cfg.hip_conv3x3 and cfg.stem_three_bf16
                and x.shape[-1] == 224 and x.shape[-2] == 224 and 0 < x.shape[0] <= 65535)

    def _prepare_h2(self, cache):
        """Per layer: conv1 (norm2 folded) and conv2 as two fp16 terms with per-output-channel scales, and the host
        constants max |norm1 scale| / max |norm1 shift| that turn the buffer's published range into a bound."""
        f = self.densenet121[0]
        h2 = {}
        for name, m in f.named_children():
            if name.startswith("denseblock"):
                blk = {}
                for lname, layer in m.items():
                    sc1, sh1, w1, b1 = cache[name][lname][0], cache[name][lname][1], cache[name][lname][2], cache[name][lname][3]
                    w2, osc = _split2h_weights(w1.view(w1.shape[0], w1.shape[1]))
                    c3w2p, c3osc = _conv3x3_weights_split2h(layer.conv2.weight, YTERMS_CHANNEL_ORDER)
                    blk[lname] = {"sc1": sc1, "sh1": sh1, "b1": b1, "w2": w2, "osc": osc, "c3osc": c3osc,
                                  "c3w2p": c3w2p, "ks": float(sc1.abs().max()),
                                  "kb": float(sh1.abs().max()),
                                  # |relu(W x + b)| <= max_o sum_c |W[o, c]| * max|x| + max|b|: the bottleneck's bound
                                  "yks": float(w1.view(w1.shape[0], -1).abs().sum(dim=1).max()), "ykb": float(b1.abs().max())}
                h2[name] = blk
            elif name.startswith("transition"):
                sc, sh = cache[name][0], cache[name][1]
                wt = m.conv.weight.detach().float()
                w2, osc = _split2h_weights(wt.view(wt.shape[0], wt.shape[1]))
                h2[name] = {"sc": sc, "sh": sh, "w2": w2, "osc": osc, "ks": float(sc.abs().max()), "kb": float(sh.abs().max())}
        cache["conv0_w2"] = _stem_weights_split2h(f.conv0.weight)
        cache["h2"] = h2
        self._mark_built(cache)
        return h2

    def _features_h2(self, x, cache):
        """-> feature map before norm5 [B, 1024, 7, 7]; every convolution on the matrix pipe with two fp16 terms per
        operand, ranges carried in range slots."""
        f = self.densenet121[0]
        lib = _lib.load()
        h2 = cache.get("h2") or self._prepare_h2(cache)
        cfg = _cfg(self)
        u8 = x.dtype == torch.uint8
        x = x.contiguous() if u8 else x.contiguous().float()
        b = x.shape[0]
        dev = x.device
        st = _stream(dev)
        blocks = [(n, m) for n, m in f.named_children() if n.startswith("denseblock")]
        trans = [n for n, _ in f.named_children() if n.startswith("transition")]
        nlayers = sum(len(m) for _, m in blocks)
        # range rows: one float per (buffer, image) -- the blocks' buffers, every layer's bottleneck scale, the input images
        ranges = torch.zeros((len(blocks) + nlayers + 1, b), dtype=torch.float32, device=dev)   # one fill per forward
        sc, sh = cache["norm0"]
        xr = ranges[len(blocks) + nlayers]                           # the range of every input image: one pass over them
        if u8:
            _lib.check(lib.mirx_range_absmax_u8(_ptr(x), 224 * 224, b, _ptr(self.input_mean), _ptr(self.input_std), _ptr(xr), st),
                       "mirx_range_absmax_u8")
        else:
            _lib.check(lib.mirx_range_absmax(_ptr(x), x[0].numel(), b, _ptr(xr), st), "mirx_range_absmax")
        w2, osc = cache["conv0_w2"]
        sides = [56 >> k for k in range(len(blocks))]
        rows = [len(blocks) + sum(len(m) for _, m in blocks[:k]) for k in range(len(blocks))]

        def stem(xs, dst):
            if u8:
                _lib.check(lib.mirx_stem_conv7_bn_relu_pool_split2h_u8_into(_ptr(xs), _ptr(self.input_mean), _ptr(self.input_std),
                                                                            _ptr(w2), _ptr(osc), _ptr(sc), _ptr(sh), xs.shape[0],
                                                                            224, 224, _ptr(dst), dst.shape[1] * dst.shape[2],
                                                                            _ptr(xr), _ptr(ranges[0]), st),
                           "mirx_stem_split2h_u8_into")
                return
            _lib.check(lib.mirx_stem_conv7_bn_relu_pool_split2h_into(_ptr(xs), _ptr(w2), _ptr(osc), _ptr(sc), _ptr(sh),
                                                                     xs.shape[0], 224, 224, _ptr(dst),
                                                                     dst.shape[1] * dst.shape[2], _ptr(xr), _ptr(ranges[0]), st),
                       "mirx_stem_split2h_into")

        def block_and_transition(k, bk, nxt):
            name, blk = blocks[k]
            # the transition's norm + relu + avgpool of every NEW channel comes out of the 3x3 launch that writes it (its pooled
            # twin: the strip kernel only, so not for launches small enough for the one-wave-per-block kernel)
            pool = None
            if (nxt is not None and cfg.pooled_twin and not (cfg.fused_small_maps and sides[k] in (14, 7))
                    and not lib.mirx_conv3x3_small_launch(bk.shape[0], sides[k])):
                t = h2[trans[k]]
                pool = (t["sc"], t["sh"], torch.empty((bk.shape[0], blk.cout, sides[k] // 2, sides[k] // 2),
                                                      dtype=torch.float32, device=dev))
            _dense_block_h2(blk, bk, sides[k], ranges[k], h2[name], ranges[rows[k]:rows[k] + len(blk)], self.conv1x1_timer,
                            cfg.fused_small_maps, pool)
            if nxt is not None:
                _transition_h2(bk, sides[k], h2[trans[k]], ranges[k], nxt, ranges[k + 1], self.conv1x1_timer,
                               pool[2] if pool else None, blk.cin)

        def new_buf(k, n):
            return torch.empty((n, blocks[k][1].cout, _plane_stride(sides[k], cfg)), dtype=torch.float32, device=dev)

        buf = new_buf(0, b)
        stem(x, buf)
        for k in range(len(blocks)):
            nxt = new_buf(k + 1, b) if k + 1 < len(blocks) else None
            block_and_transition(k, buf, nxt)
            if nxt is not None:
                buf = nxt
        side = sides[-1]
        buf = buf.view(b, buf.shape[1], side, side)                 # the 7 x 7 planes are packed
        self.__dict__["_mirx_last_ranges"] = ranges            # kept for diagnostics (tools/h2_state_probe.py)
        return buf

    def _head_fused(self, fmap, normalize, cache=None):
        lib = _lib.load()
        sc, sh = (cache or self._cache())["norm5"]
        fmap = fmap.contiguous()
        b, c, h, w = fmap.shape
        out = torch.empty((b, c), dtype=torch.float32, device=fmap.device)
        _lib.check(lib.mirx_bn_relu_gap_l2norm(_ptr(fmap), _ptr(sc), _ptr(sh), b, c, h * w,
                                               1 if normalize else 0, _ptr(out), _stream(fmap.device)),
                   "mirx_bn_relu_gap_l2norm")
        return out

    def forward(self, x):
        fused = x.is_cuda and not self.training and not torch.is_grad_enabled()
        if fused:
            with torch.cuda.device(x.device):
                cache = self._cache()                  # ONE validity check of the folded weights per forward
                fmap = self._features_fused(x, cache)
                plain_head = self.fc is None and self.classification_head is None
                x = self._head_fused(fmap, normalize=plain_head, cache=cache)
                if plain_head:
                    return x                       # already unit-norm (model.py:83)
        else:
            x = self.forward_eager(self.normalize_uint8(x) if x.dtype == torch.uint8 else x)
        if self.fc:
            x = _linear_auto(self.fc, x)         # [HIP] three-bf16-term Linear on the inference path: no library GEMM in a forward
        if self.classification_head is not None:
            return {"embedding": F.normalize(x, dim=1), "logits": self.classification_head(x)}
        return F.normalize(x, dim=1)


# =================================================================================================
# ConvNeXtV2-base (reference model.py:87-117 around timm 'convnextv2_base.fcmae_ft_in22k_in1k_384',
# num_classes=0).  timm is not a dependency: the module tree below reproduces its parameter names
# (stem.0/1, stages.N.downsample.0/1, stages.N.blocks.M.{conv_dw,norm,mlp.fc1,mlp.grn,mlp.fc2},
# head.norm) so reference checkpoints (`convnext.*`, `fc.*`) load unchanged.
# =================================================================================================
CNX_DEPTHS = (3, 3, 27, 3)
CNX_DIMS = (128, 256, 512, 1024)
CNX_EPS = 1e-6


class _LayerNorm2d(nn.LayerNorm):
    """LayerNorm over the channel axis of an NCHW tensor (timm LayerNorm2d)."""

    def forward(self, x):
        return F.layer_norm(x.permute(0, 2, 3, 1), self.normalized_shape, self.weight, self.bias,
                            self.eps).permute(0, 3, 1, 2)


class _GRN(nn.Module):
    """Global response normalisation on NHWC (timm GlobalResponseNorm, channels_last)."""

    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(dim))
        self.bias = nn.Parameter(torch.zeros(dim))

    def forward(self, x):
        g = torch.linalg.vector_norm(x, ord=2, dim=(1, 2), keepdim=True)
        n = g / (g.mean(dim=-1, keepdim=True) + 1e-6)
        return x + torch.addcmul(self.bias, self.weight, x * n)


class _CnxMlp(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.fc1 = nn.Linear(dim, 4 * dim)
        self.act = nn.GELU()
        self.grn = _GRN(4 * dim)
        self.fc2 = nn.Linear(4 * dim, dim)

    def forward(self, x):
        if _linear_s3_ok(self.fc1, x) and _linear_s3_ok(self.fc2, x):
            return _linear_s3(self.fc2, self.grn(_linear_s3(self.fc1, x, act=1)))      # bias + GELU in the epilogue
        return self.fc2(self.grn(self.act(self.fc1(x))))


class _CnxBlock(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv_dw = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=CNX_EPS)
        self.mlp = _CnxMlp(dim)

    def _fc2_bias_with_grn_shift(self):
        fc2, grn = self.mlp.fc2, self.mlp.grn
        key = (fc2.weight.data_ptr(), fc2.weight._version, grn.bias._version,
               None if fc2.bias is None else fc2.bias._version, fc2.weight.device)
        cached = getattr(self, "_mirx_bias2", None)
        if cached is None or cached[0] != key:
            bias = (fc2.weight.detach().double() @ grn.bias.detach().double()).float()
            if fc2.bias is not None:
                bias = bias + fc2.bias.detach()
            cached = (key, bias.contiguous())
            self._mirx_bias2 = cached
        return cached[1]

    def _forward_nhwc(self, t, b, h, w):
        """[HIP] The block on a channels-last residual stream t [b * h * w, c] (ConvNeXtV2 fast path): depthwise 7x7 without
        LDS (a lane = a channel), LayerNorm, fc1 + GELU, GRN as a norm pass + a scale folded into fc2's staging, fc2 with the
        skip added in its row-major epilogue.  -> the new stream (a fresh tensor)."""
        lib = _lib.load()
        mlp, c, dev = self.mlp, t.shape[-1], t.device
        wk = self.conv_dw.weight
        key = _tkey(wk)
        cached = getattr(self, "_mirx_dw_t", None)
        if cached is None or cached[0] != key:
            cached = (key, wk.detach().reshape(c, 49).t().contiguous())          # [49][c]: a lane reads its channel's taps coalesced
            self._mirx_dw_t = cached
        y = torch.empty_like(t)
        with torch.cuda.device(dev):
            _lib.check(lib.mirx_dwconv7x7_nhwc(_ptr(t), _ptr(cached[1]), _ptr(self.conv_dw.bias.detach()), b, c, h, w, _ptr(y),
                                               _stream(dev)), "mirx_dwconv7x7_nhwc")
        yn = _layernorm(self.norm, y)
        bn = _layernorm_bound(self.norm)
        c4 = mlp.fc1.out_features
        gx = torch.empty((b, c4), dtype=torch.float32, device=dev)
        fused_norm = h * w >= 128 and _cfg(self).cnx_grn_in_fc1
        if fused_norm:
            # fc1 + GELU returns the GRN partial sums of its own output tiles: the norm pass over the 4C-wide map disappears
            w2f, wsf = _linear_h2_weights(mlp.fc1)
            xs = _terms_scale(bn)
            hid = torch.empty((b * h * w, c4), dtype=torch.float32, device=dev)   # [b * h * w, 4c]
            parts = torch.empty(((b * h * w + 127) // 128) * 2 * c4, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.mirx_linear_split2h_gelu_grn(_ptr(yn), b, h * w, c, _ptr(w2f), _ptr(mlp.fc1.bias.detach()), c4, xs,
                                                            1.0 / (xs * wsf), _ptr(hid), _ptr(parts), _ptr(gx), _stream(dev)),
                           "mirx_linear_split2h_gelu_grn")
        else:
            hid = _linear_h2(mlp.fc1, yn, bn, act=1)                              # [b * h * w, 4c]
        scale = torch.empty_like(gx)
        smax = torch.zeros(1, dtype=torch.float32, device=dev)
        out = torch.empty_like(t)
        w2, ws = _linear_h2_weights(mlp.fc2)
        hb = _linear_out_bound(self.norm, mlp.fc1)                                # |gelu(fc1(LN(.)))| <= hb
        with torch.cuda.device(dev):
            st = _stream(dev)
            if not fused_norm:
                _lib.check(lib.mirx_grn_norm_nhwc(_ptr(hid), b, h * w, c4, _ptr(gx), st), "mirx_grn_norm_nhwc")
            _lib.check(lib.mirx_grn_scale(_ptr(gx), _ptr(mlp.grn.weight.detach().reshape(-1).contiguous()), b, c4, 1e-6,
                                          _ptr(scale), _ptr(smax), st), "mirx_grn_scale")
            _lib.check(lib.mirx_linear_split2h_grn_rows(_ptr(hid), b, h * w, c4, _ptr(w2), _ptr(self._fc2_bias_with_grn_shift()), c,
                                                        _ptr(t), _ptr(scale), float(hb), _ptr(smax), 1.0 / ws, _ptr(out), st),
                       "mirx_linear_split2h_grn_rows")
        return out

    def _nhwc_ok(self):
        mlp = self.mlp
        bn = _layernorm_bound(self.norm)
        hb = _linear_out_bound(self.norm, mlp.fc1)
        probe = self.conv_dw.weight
        return (_cfg(self).grn_scale_kernel and probe.is_cuda and self.conv_dw.bias is not None and mlp.fc1.bias is not None
                and _linear_h2_ok(mlp.fc1, probe, bn) and _linear_h2_ok(mlp.fc2, probe, hb))

    def forward(self, x):
        if x.is_cuda and not torch.is_grad_enabled() and x.dtype == torch.float32:
            # MI355X path: depthwise 7x7 + the NCHW->NHWC permute in one HIP pass
            lib = _lib.load()
            xc = x.contiguous()
            b, c, h, w = xc.shape
            y = torch.empty((b, h, w, c), dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                _lib.check(lib.mirx_dwconv7x7_nchw_to_nhwc(_ptr(xc), _ptr(self.conv_dw.weight.detach().contiguous()),
                                                           _ptr(self.conv_dw.bias.detach().contiguous()), b, c, h, w,
                                                           _ptr(y), _stream(x.device)), "mirx_dwconv7x7")
        else:
            y = self.conv_dw(x).permute(0, 2, 3, 1)
        mlp = self.mlp
        if (x.is_cuda and y.is_contiguous() and x.shape[0] <= 65535 and _linear_s3_ok(mlp.fc1, y)
                and _linear_s3_ok(mlp.fc2, y)):
            # MI355X path: fc1 + GELU in one MFMA launch, GRN = one norm pass + a scale folded into fc2's staging,
            # fc2 written back NCHW with the skip added in its epilogue
            lib = _lib.load()
            b, c, h, w = x.shape
            yn = _layernorm(self.norm, y)
            bn = _layernorm_bound(self.norm)                               # fc1 reads a LayerNorm output
            hid = (_linear_h2(mlp.fc1, yn, bn, act=1) if _linear_h2_ok(mlp.fc1, yn, bn)
                   else _linear_s3(mlp.fc1, yn, act=1))                    # [b, h, w, 4c]
            c4 = hid.shape[-1]
            gx = torch.empty((b, c4), dtype=torch.float32, device=x.device)
            out = torch.empty_like(xc)
            with torch.cuda.device(x.device):
                st = _stream(x.device)
                _lib.check(lib.mirx_grn_norm_nhwc(_ptr(hid), b, h * w, c4, _ptr(gx), st), "mirx_grn_norm_nhwc")
                if _cfg(self).grn_scale_kernel:
                    scale = torch.empty_like(gx)
                    smax = torch.zeros(1, dtype=torch.float32, device=x.device)
                    _lib.check(lib.mirx_grn_scale(_ptr(gx), _ptr(mlp.grn.weight.detach().reshape(-1).contiguous()), b, c4, 1e-6,
                                                  _ptr(scale), _ptr(smax), st), "mirx_grn_scale")
                else:                                                      # the ATen expression of the same vector (A/B)
                    scale = torch.addcmul(torch.ones_like(gx), mlp.grn.weight.detach(),
                                          gx / (gx.mean(dim=-1, keepdim=True) + 1e-6))
                    smax = scale.abs().amax().reshape(1)
                # GRN apply folded into the second Linear: the scale multiplies x while it is staged, the shift
                # is constant per feature, so W (x s + b) + bias = W (x s) + (bias + W b)
                hb = _linear_out_bound(self.norm, mlp.fc1)                 # |gelu(fc1(LN(.)))| <= |fc1(LN(.))| <= hb
                if _linear_h2_ok(mlp.fc2, hid, hb):
                    # two fp16 terms: |hid * scale| <= hb * max |scale|; the second factor is data, so it stays on the
                    # device (one scalar) and the kernel derives its staging scale from it
                    w2, ws = _linear_h2_weights(mlp.fc2)
                    _lib.check(lib.mirx_linear_split2h_nchw(_ptr(hid), b, h * w, c4, _ptr(w2),
                                                            _ptr(self._fc2_bias_with_grn_shift()), c, _ptr(xc), _ptr(scale),
                                                            float(hb), _ptr(smax), 1.0 / ws, _ptr(out), st),
                               "mirx_linear_split2h_nchw")
                else:
                    _lib.check(lib.mirx_linear_split3_nchw(_ptr(hid), b, h * w, c4, _ptr(_linear_w3(mlp.fc2)),
                                                           _ptr(self._fc2_bias_with_grn_shift()), c, _ptr(xc), _ptr(scale),
                                                           _ptr(out), st), "mirx_linear_split3_nchw")
            return out
        y = self.mlp(self.norm(y))
        return y.permute(0, 3, 1, 2) + x


class _CnxStage(nn.Module):
    def __init__(self, cin, cout, depth, downsample):
        super().__init__()
        if downsample:
            self.downsample = nn.Sequential(_LayerNorm2d(cin, eps=CNX_EPS), nn.Conv2d(cin, cout, kernel_size=2, stride=2))
        else:
            self.downsample = nn.Identity()
        self.blocks = nn.Sequential(*[_CnxBlock(cout) for _ in range(depth)])

    def forward(self, x):
        ds = self.downsample
        if (isinstance(ds, nn.Sequential) and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()
                and x.shape[0] <= 65535 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and _cfg(self).linear_three_bf16):
            # MI355X path: LayerNorm2d + the NCHW -> patch-row gather in one HIP pass, the 2x2/2 conv as an MFMA Linear
            # that writes the next stage's NCHW map
            x = _conv_patch_tokens(ds[1], x, ln2d=ds[0], nchw_out=True)
        else:
            x = ds(x)
        return self.blocks(x)


class _CnxHead(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.norm = _LayerNorm2d(dim, eps=CNX_EPS)

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled():
            return _layernorm(self.norm, x.mean(dim=(2, 3)))
        return torch.flatten(self.norm(x.mean(dim=(2, 3), keepdim=True)), 1)


class _ConvNeXtV2Backbone(nn.Module):
    """timm-shaped backbone (num_classes=0): forward(x) -> pooled, head-normalised features."""

    def __init__(self):
        super().__init__()
        self.num_features = CNX_DIMS[-1]
        self.stem = nn.Sequential(nn.Conv2d(3, CNX_DIMS[0], kernel_size=4, stride=4), _LayerNorm2d(CNX_DIMS[0], eps=CNX_EPS))
        stages, cin = [], CNX_DIMS[0]
        for i, (d, c) in enumerate(zip(CNX_DEPTHS, CNX_DIMS)):
            stages.append(_CnxStage(cin, c, d, downsample=i > 0))
            cin = c
        self.stages = nn.Sequential(*stages)
        self.head = _CnxHead(cin)
        for m in self.modules():           # timm's init: trunc_normal(std=.02) weights, zero biases
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def _stem(self, x):
        if (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and x.shape[0] <= 65535 and _cfg(self).linear_three_bf16
                and x.shape[2] % 4 == 0 and x.shape[3] % 4 == 0):
            b, _, h, w = x.shape
            tok = _conv_patch_tokens(self.stem[0], x)                          # [b * h/4 * w/4, 128]
            return _layernorm(self.stem[1], tok, tokens_per_image=(h // 4) * (w // 4)).view(b, -1, h // 4, w // 4)
        return self.stem(x)

    def _forward_nhwc(self, x):
        """[HIP] The whole backbone with a channels-last residual stream [b * h * w, c] (no NCHW map between the patch embedding
        and the pooled head): stem = patch gather + MFMA Linear + LayerNorm rows; downsample = LayerNorm written straight into
        2 x 2 patch rows + MFMA Linear (weight in (ky, kx, c) order); blocks = _CnxBlock._forward_nhwc."""
        lib = _lib.load()
        b, _, h, w = x.shape
        dev = x.device
        t = _layernorm(self.stem[1], _conv_patch_tokens(self.stem[0], x))                      # [b * h/4 * w/4, 128]
        h, w = h // 4, w // 4
        for stage in self.stages:
            ds = stage.downsample
            if isinstance(ds, nn.Sequential):
                ln, conv = ds[0], ds[1]
                pl = conv.__dict__.get("_mirx_as_linear_cl")
                if pl is None:
                    pl = _ConvAsLinear(conv, channels_last=True)
                    conv.__dict__["_mirx_as_linear_cl"] = pl
                pl.__dict__["_mirx_cfg"] = _cfg(conv)
                pl.refresh()
                c = t.shape[-1]
                rows = torch.empty((b * (h // 2) * (w // 2), 4 * c), dtype=torch.float32, device=dev)
                with torch.cuda.device(dev):
                    _lib.check(lib.mirx_layernorm_patch2_nhwc(_ptr(t), b, h, w, c, _ptr(ln.weight.detach()), _ptr(ln.bias.detach()),
                                                              float(ln.eps), _ptr(rows), _stream(dev)), "mirx_layernorm_patch2_nhwc")
                h, w = h // 2, w // 2
                bound = _layernorm_bound(ln)
                t = _linear_h2(pl, rows, bound) if _linear_h2_ok(pl, rows, bound) else _linear_s3(pl, rows)
            for blk in stage.blocks:
                t = blk._forward_nhwc(t, b, h, w)
        pooled = t.view(b, h * w, t.shape[-1]).mean(dim=1)                                    # global average pool
        return _layernorm(self.head.norm, pooled)

    def _nhwc_ok(self, x):
        if not (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and x.dim() == 4 and x.shape[0] <= 65535):
            return False
        cfg = _cfg(self)
        if not (cfg.cnx_channels_last and cfg.linear_three_bf16 and cfg.linear_two_fp16):
            return False
        h, w = x.shape[2], x.shape[3]
        if h % 32 or w % 32:                                  # stem / 4, three downsamples / 2
            return False
        return all(blk._nhwc_ok() for stage in self.stages for blk in stage.blocks)

    def forward(self, x):
        if self._nhwc_ok(x):
            return self._forward_nhwc(x.contiguous())
        return self.head(self.stages(self._stem(x)))


class ConvNeXtV2(_Configurable, nn.Module):
    """Reference model.py:87-117: `convnext` backbone, optional `fc`, unit-norm output."""

    def __init__(self, pretrained=False, embedding_dim=None, weights=None):
        super().__init__()
        if pretrained and weights is None:
            raise RuntimeError("pretrained=True needs a download in the reference (model.py:96-100); "
                               "pass weights=<state dict or path> instead")
        self.convnext = _ConvNeXtV2Backbone()
        in_features = self.convnext.num_features
        self.fc = nn.Linear(in_features, embedding_dim) if embedding_dim else None
        if weights is not None:
            sd = torch.load(weights, map_location="cpu") if isinstance(weights, str) else weights
            for key in ("state-dict", "state_dict"):
                if isinstance(sd, dict) and key in sd:
                    sd = sd[key]
            self.load_state_dict(sd, strict=False)

    def forward(self, x):
        x = self.convnext(x)
        x = torch.flatten(x, 1)
        if self.fc:
            x = _linear_auto(self.fc, x)         # [HIP] three-bf16-term Linear on the inference path: no library GEMM in a forward
        if x.is_cuda and not torch.is_grad_enabled() and x.dtype == torch.float32 and x.is_contiguous():
            from .index import l2_normalize_
            return l2_normalize_(x)                # HIP F.normalize (model.py:116)
        return F.normalize(x, dim=1)


# =================================================================================================
# DINOv2 ViT-B/14 (reference model.py:448-494 and nih_multilabel_retrieval.py:170-221 around timm
# 'vit_base_patch14_dinov2.lvd142m', num_classes=0).  Module tree reproduces timm's parameter names
# (cls_token, pos_embed, patch_embed.proj, blocks.N.{norm1,attn.qkv,attn.proj,ls1.gamma,norm2,
# mlp.fc1,mlp.fc2,ls2.gamma}, norm) so `backbone.*` checkpoints load unchanged.
# =================================================================================================
class _LayerScale(nn.Module):
    def __init__(self, dim, init=1e-5):
        super().__init__()
        self.gamma = nn.Parameter(init * torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class _VitAttention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        b, n, c = x.shape
        dh = c // self.num_heads
        qkv = self.qkv(x)                                                # [b, n, 3, heads, dh] packed
        if x.is_cuda and dh in (32, 64, 72, 96) and qkv.dtype == torch.float32 and not torch.is_grad_enabled():
            # MI355X inference path: fp32 MFMA flash attention straight on the packed projection,
            # output already [b, n, heads * dh] (include/mirx.h: mirx_attention_qkv_f32)
            qkv = qkv.contiguous()
            a = torch.empty((b, n, c), dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                lib = _lib.load()
                att = lib.mirx_attention_qkv_f32_split3 if _cfg(self).attention_three_bf16 else lib.mirx_attention_qkv_f32
                _lib.check(att(_ptr(qkv), b, n, self.num_heads, dh, float(dh) ** -0.5, _ptr(a), _stream(x.device)),
                           "mirx_attention_qkv_f32")
            return self.proj(a)
        qkv = qkv.reshape(b, n, 3, self.num_heads, dh).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])      # softmax(q k^T / sqrt(d)) v
        return self.proj(a.transpose(1, 2).reshape(b, n, c))


class _VitMlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _VitBlock(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _VitAttention(dim, heads)
        self.ls1 = _LayerScale(dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _VitMlp(dim, 4 * dim)
        self.ls2 = _LayerScale(dim)

    def forward(self, x):
        at = self.attn
        if (x.dim() == 3 and x.shape[-1] // at.num_heads == 64 and _linear_s3_ok(at.qkv, x)
                and _linear_s3_ok(at.proj, x) and _linear_s3_ok(self.mlp.fc1, x) and _linear_s3_ok(self.mlp.fc2, x)):
            # MI355X inference path: 4 split-3 MFMA Linear launches (bias / GELU / LayerScale + skip in their
            # epilogues) + the flash-attention kernel + 2 LayerNorms per block
            b, n, c = x.shape
            x = x.contiguous()
            # the two Linears fed by a LayerNorm have a provable input bound: two fp16 terms (3 MFMAs per product)
            b1, b2 = _layernorm_bound(self.norm1), _layernorm_bound(self.norm2)
            # attention output = softmax-weighted average of V rows: bounded like the V part of the qkv projection; |gelu(v)| <= |v|
            ba, bh = _linear_out_bound(self.norm1, at.qkv, slice(2 * c, 3 * c)), _linear_out_bound(self.norm2, self.mlp.fc1)
            # big batches: every Linear on the DMA-fed kernel, its input handed over as terms rows by the producer
            terms = _linear_terms_ok(self, b * n, (at.qkv, at.proj, self.mlp.fc1, self.mlp.fc2), (b1, b2, ba, bh))
            if terms:
                h1t, s1 = _layernorm_terms(self.norm1, x, b1)
                qkv = _linear_terms(at.qkv, h1t, s1, (b, n))
            else:
                h1 = _layernorm(self.norm1, x)
                qkv = _linear_h2(at.qkv, h1, b1) if _linear_h2_ok(at.qkv, h1, b1) else _linear_s3(at.qkv, h1)
            bqk = _linear_out_bound(self.norm1, at.qkv, slice(0, 2 * c))
            bv = _linear_out_bound(self.norm1, at.qkv, slice(2 * c, 3 * c))
            att = None
            if terms and _cfg(self).attention_two_fp16 and 0.0 < bqk < 3.0e4:
                # the attention kernel hands its output to the projection as terms rows too (|out| <= ba)
                att, sa = (torch.empty if c % 32 == 0 else torch.zeros)((b * n, (c + 31) // 32 * 64), dtype=torch.float16, device=x.device), _terms_scale(ba)
                with torch.cuda.device(x.device):
                    _lib.check(_lib.load().mirx_attention_qkv_f32_split2h_terms(_ptr(qkv), b, n, at.num_heads, 64, 0.125, bqk, bv, sa,
                                                                                _ptr(att), _stream(x.device)),
                               "mirx_attention_qkv_f32_split2h_terms")
            a = torch.empty((b, n, c), dtype=torch.float32, device=x.device) if att is None else None
            with torch.cuda.device(x.device):
                lib = _lib.load()
                if att is not None:
                    pass
                elif _cfg(self).attention_two_fp16 and 0.0 < bqk < 3.0e4 and 0.0 < bv < 3.0e4:
                    # q, k, v are outputs of a LayerNorm-fed Linear: provable bounds -> two fp16 terms per operand
                    _lib.check(lib.mirx_attention_qkv_f32_split2h(_ptr(qkv), b, n, at.num_heads, 64, 0.125, bqk, bv, _ptr(a),
                                                                  _stream(x.device)), "mirx_attention_qkv_f32_split2h")
                else:
                    att = lib.mirx_attention_qkv_f32_split3 if _cfg(self).attention_three_bf16 else lib.mirx_attention_qkv_f32
                    _lib.check(att(_ptr(qkv), b, n, at.num_heads, 64, 0.125, _ptr(a), _stream(x.device)),
                               "mirx_attention_qkv_f32")
            if terms:
                if att is None:
                    att, sa = _rows_to_terms(a, ba)
                x = _linear_terms(at.proj, att, sa, (b, n), res=x, gamma=self.ls1.gamma)
                h2t, s2 = _layernorm_terms(self.norm2, x, b2)
                hidt, sh = _linear_terms(self.mlp.fc1, h2t, s2, (b, n), act=1, terms_bound=bh)
                return _linear_terms(self.mlp.fc2, hidt, sh, (b, n), res=x, gamma=self.ls2.gamma, out=x)
            x = (_linear_h2(at.proj, a, ba, res=x, gamma=self.ls1.gamma) if _linear_h2_ok(at.proj, a, ba)
                 else _linear_s3(at.proj, a, res=x, gamma=self.ls1.gamma))
            h2 = _layernorm(self.norm2, x)
            hid = (_linear_h2(self.mlp.fc1, h2, b2, act=1) if _linear_h2_ok(self.mlp.fc1, h2, b2)
                   else _linear_s3(self.mlp.fc1, h2, act=1))
            bh = _linear_out_bound(self.norm2, self.mlp.fc1)                 # |gelu(v)| <= |v|
            if _linear_h2_ok(self.mlp.fc2, hid, bh):
                return _linear_h2(self.mlp.fc2, hid, bh, res=x, gamma=self.ls2.gamma, out=x)
            return _linear_s3(self.mlp.fc2, hid, res=x, gamma=self.ls2.gamma, out=x)
        x = x + self.ls1(self.attn(self.norm1(x)))
        return x + self.ls2(self.mlp(self.norm2(x)))


class _PatchEmbed(nn.Module):
    def __init__(self, dim, patch):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=patch, stride=patch)

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and x.shape[0] <= 65535 and _cfg(self).linear_three_bf16:
            # MI355X path: the stride-14 patch convolution as patch gather + MFMA Linear (no library convolution)
            return _conv_patch_tokens(self.proj, x).view(x.shape[0], -1, self.proj.out_channels)
        return self.proj(x).flatten(2).transpose(1, 2)


class _Dinov2Backbone(nn.Module):
    """timm VisionTransformer surface used by the reference: forward(x) -> CLS feature [B,C],
    forward_features(x) -> tokens [B,1+N,C], .blocks, .norm, .num_features."""

    def __init__(self, img_size=518, patch=14, dim=768, depth=12, heads=12):
        super().__init__()
        self.num_features = dim
        self.patch_size = patch
        self.grid = img_size // patch
        self.patch_embed = _PatchEmbed(dim, patch)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(0.02 * torch.randn(1, 1 + self.grid * self.grid, dim))
        self.blocks = nn.Sequential(*[_VitBlock(dim, heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def _pos(self, gh, gw):
        if gh == self.grid and gw == self.grid:
            return self.pos_embed
        # other input sizes: bicubic resampling of the patch grid (timm resample_abs_pos_embed)
        cls, grid = self.pos_embed[:, :1], self.pos_embed[:, 1:]
        grid = grid.reshape(1, self.grid, self.grid, -1).permute(0, 3, 1, 2)
        grid = F.interpolate(grid, size=(gh, gw), mode="bicubic", antialias=True, align_corners=False)
        return torch.cat([cls, grid.permute(0, 2, 3, 1).reshape(1, gh * gw, -1)], dim=1)

    def forward_features(self, x):
        gh, gw = x.shape[-2] // self.patch_size, x.shape[-1] // self.patch_size
        x = self.patch_embed(x)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1) + self._pos(gh, gw)
        return _layernorm(self.norm, self.blocks(x))

    def forward(self, x):
        return self.forward_features(x)[:, 0]


def _normalize_rows(x):
    if x.is_cuda and not torch.is_grad_enabled() and x.dtype == torch.float32:
        from .index import l2_normalize_
        return l2_normalize_(x.contiguous().clone())
    return F.normalize(x, dim=1)


class DinoV2(_Configurable, nn.Module):
    """Reference model.py:448-494 (`backbone`, `fc`; the last `unfreeze_blocks` blocks + final norm
    trainable, the rest frozen)."""

    def __init__(self, model_name="vit_base_patch14_dinov2.lvd142m", pretrained=False, embedding_dim=None,
                 unfreeze_blocks=3, weights=None, img_size=518):
        super().__init__()
        if pretrained and weights is None:
            raise RuntimeError("pretrained=True needs a download in the reference (model.py:459-463); "
                               "pass weights=<state dict or path> instead")
        if model_name != "vit_base_patch14_dinov2.lvd142m":
            raise ValueError(f"Unknown DINOv2 backbone: {model_name}")
        self.backbone = _Dinov2Backbone(img_size=img_size)
        for p in self.backbone.parameters():
            p.requires_grad = False
        nb = max(0, min(unfreeze_blocks, len(self.backbone.blocks)))
        if nb > 0:
            for blk in list(self.backbone.blocks)[-nb:]:
                for p in blk.parameters():
                    p.requires_grad = True
        for p in self.backbone.norm.parameters():
            p.requires_grad = True
        self.fc = nn.Linear(self.backbone.num_features, embedding_dim) if embedding_dim else None
        if weights is not None:
            sd = torch.load(weights, map_location="cpu") if isinstance(weights, str) else weights
            for key in ("state-dict", "state_dict"):
                if isinstance(sd, dict) and key in sd:
                    sd = sd[key]
            self.load_state_dict(sd, strict=False)

    def forward(self, x):
        x = torch.flatten(self.backbone(x), 1)
        if self.fc:
            x = _linear_auto(self.fc, x)         # [HIP] three-bf16-term Linear on the inference path: no library GEMM in a forward
        return _normalize_rows(x)


class DINOv2MultiLabelRetrievalModel(_Configurable, nn.Module):
    """Reference nih_multilabel_retrieval.py:170-221: dict with cls_embedding / projection /
    embedding (unit norm, 256-d) / logits."""

    def __init__(self, num_labels=14, backbone_name="vit_base_patch14_dinov2.lvd142m", pretrained=False, img_size=518):
        super().__init__()
        if pretrained:
            raise RuntimeError("pretrained=True needs a download in the reference; load a state dict instead")
        if backbone_name != "vit_base_patch14_dinov2.lvd142m":
            raise ValueError(f"Unknown DINOv2 backbone: {backbone_name}")
        self.backbone = _Dinov2Backbone(img_size=img_size)
        self.projection_head = nn.Sequential(nn.Linear(self.backbone.num_features, 512), nn.GELU(), nn.Linear(512, 256))
        self.classification_head = nn.Linear(256, num_labels)

    def forward(self, images):
        cls_embedding = self.backbone.forward_features(images)[:, 0]
        projection = self.projection_head(cls_embedding)
        return {"cls_embedding": cls_embedding, "projection": projection,
                "embedding": _normalize_rows(projection), "logits": self.classification_head(projection)}


class ConvNeXtV2MultiLabelRetrievalModel(_Configurable, nn.Module):
    """Reference nih_multilabel_retrieval.py:224-257."""

    def __init__(self, num_labels=14, backbone_name="convnextv2_base.fcmae_ft_in22k_in1k_384", pretrained=False):
        super().__init__()
        if pretrained:
            raise RuntimeError("pretrained=True needs a download in the reference; load a state dict instead")
        self.backbone = _ConvNeXtV2Backbone()
        self.projection_head = nn.Sequential(nn.Linear(self.backbone.num_features, 512), nn.GELU(), nn.Linear(512, 256))
        self.classification_head = nn.Linear(256, num_labels)

    def forward(self, images):
        features = self.backbone(images)
        projection = self.projection_head(features)
        return {"backbone_embedding": features, "projection": projection,
                "embedding": _normalize_rows(projection), "logits": self.classification_head(projection)}


# =================================================================================================
# MedSigLIP (reference model.py:536-634): SigLIP so400m vision tower -> pooler_output ->
# Linear(h,512)-LayerNorm-ReLU-Linear(512,embed_dim) -> L2 normalise.  The reference takes the tower
# from `AutoModel.from_pretrained("google/medsiglip-448").vision_model` (a download); here the same
# transformers class is built from a LOCAL config, so `backbone.*` / `projection.*` checkpoints
# load unchanged and nothing is fetched.
# =================================================================================================
class MedSigLIP(_Configurable, nn.Module):
    """Reference model.py:536-634: `backbone` (the SigLIP vision tower), `projection`, unit-norm output."""

    def __init__(self, model_name="google/medsiglip-448", embed_dim=512, unfreeze_layers=2, vision_config=None,
                 weights=None):
        super().__init__()
        from .siglip import MEDSIGLIP_VISION, SiglipVisionTower
        self.backbone = SiglipVisionTower(**(vision_config or MEDSIGLIP_VISION))
        for p in self.backbone.parameters():
            p.requires_grad = False
        if unfreeze_layers > 0:
            for layer in self.backbone.encoder.layers[-unfreeze_layers:]:
                for p in layer.parameters():
                    p.requires_grad = True
            for p in self.backbone.post_layernorm.parameters():
                p.requires_grad = True
        hidden = self.backbone.config.hidden_size
        self.projection = nn.Sequential(nn.Linear(hidden, 512), nn.LayerNorm(512), nn.ReLU(), nn.Linear(512, embed_dim))
        if weights is not None:
            sd = torch.load(weights, map_location="cpu") if isinstance(weights, str) else weights
            for key in ("state-dict", "state_dict"):
                if isinstance(sd, dict) and key in sd:
                    sd = sd[key]
            self.load_state_dict(sd, strict=False)

    def ensure_eager_attention(self):
        """Kept for callers (milvus_retrieval.py:172); attention maps always come from the explicit softmax path."""
        self.backbone.config._attn_implementation = "eager"

    def verify_attention_output(self, device="cuda"):
        self.eval()
        size = self.backbone.config.image_size
        with torch.no_grad():
            out = self.backbone(pixel_values=torch.randn(1, 3, size, size, device=device), output_attentions=True,
                                return_dict=True)
        return out.attentions is not None and len(out.attentions) > 0 and out.attentions[0].numel() > 0

    def forward(self, x):
        features = self.backbone(pixel_values=x).pooler_output
        p = self.projection
        h = torch.relu(_layernorm(p[1], _linear_auto(p[0], features)))
        return _normalize_rows(_linear_auto(p[3], h))


def build_model(model_type, embedding_dim=None, **kw):
    """Factory in the spirit of milvus_retrieval.py:143-162 (unknown type -> ValueError)."""
    if model_type == "densenet121":
        return DenseNet121(embedding_dim=embedding_dim, **kw), 224
    if model_type == "convnextv2":
        return ConvNeXtV2(embedding_dim=embedding_dim, **kw), 384
    if model_type == "dinov2":
        return DinoV2(embedding_dim=embedding_dim, **kw), 518
    if model_type == "medsiglip":
        return MedSigLIP(embed_dim=embedding_dim if embedding_dim is not None else 512, **kw), 448
    if model_type in ("resnet50", "convnextv2_sra"):
        # named in the reference's MODEL_CONFIGS (collection names) and kept there for compatibility, but not on the hot path
        # SURVEY section 8 scopes (DenseNet-121, ConvNeXtV2, DINOv2, MedSigLIP): no MI355X-native forward exists for them
        raise ValueError(f"Unknown model type: {model_type} (outside the accelerated path: build it with the reference's own "
                         f"model.py and feed its embeddings to MilvusRetriever / FlatIndex)")
    raise ValueError(f"Unknown model type: {model_type}")
