"""Sustained shader clock under each hot kernel: a one-wave probe (tools/clock_probe.hip) spins on its own stream for
a few milliseconds while the kernel under test is launched back to back on the default stream.
Usage on the GPU box:  hipcc --offload-arch=gfx950 -shared -fPIC tools/clock_probe.hip -o gpurun_out/libprobe.so &&
python tools/clock_probe.py gpurun_out/libprobe.so"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirx import _lib                                                   # noqa: E402
from mirx.index import FlatIndex                                        # noqa: E402
from mirx.model import DenseNet121, _split3_weights, _winograd_weights  # noqa: E402


def main():
    probe = ctypes.CDLL(sys.argv[1])
    probe.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p]
    lib = _lib.load()
    dev = torch.device("cuda:0")
    vp = lambda t: ctypes.c_void_p(t.data_ptr())                        # noqa: E731
    side = torch.cuda.Stream()
    out = torch.zeros(2, dtype=torch.int64, device=dev)

    def measure(name, fn, ms=8):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        for _ in range(200):                                            # keep the default stream busy for >> ms
            fn()
            if _ == 2:
                probe.clock_probe_launch(vp(out), ms * 1000, ctypes.c_void_p(side.cuda_stream))
        torch.cuda.synchronize()
        c, r = (int(v) for v in out.cpu())
        print(f"{name:44s} {100.0 * c / max(r, 1):7.0f} MHz  (probe window {r / 100.0:.0f} us)", flush=True)

    measure("idle (probe alone)", lambda: None, ms=2)
    if "--h2" in sys.argv:                                              # round 2: the two-fp16-term DenseNet kernels only
        from mirx.model import _conv3x3_weights_split2h, _split2h_weights
        g = torch.Generator(device=dev).manual_seed(0)
        w2, osc = _conv3x3_weights_split2h(torch.randn(32, 128, 3, 3, generator=g, device=dev) * 0.05)
        yinv = torch.full((1,), 2.0 ** -7, device=dev)
        for sd in (56, 28):
            yt = (torch.randn(1024, 8, 2, sd * sd, 16, generator=g, device=dev) * 100).half()
            oo = torch.empty(1024, 32, sd, sd, device=dev)
            measure(f"k_conv3x3_d2p<{sd}> (1024 images)", lambda: _lib.check(lib.mirx_conv3x3_direct_terms_nchw(
                vp(yt), vp(w2), vp(osc), 1024, sd, vp(oo), 32 * sd * sd, vp(yinv), None, 0, None), "c"))
        cin, hw = 256, 3136
        xb = torch.randn(1024, cin, hw, generator=g, device=dev)
        w1, o1 = _split2h_weights(torch.randn(128, cin, generator=g, device=dev) / 16)
        sc, sh, bias = torch.ones(cin, device=dev), torch.zeros(cin, device=dev), torch.zeros(128, device=dev)
        rng = torch.zeros(64, device=dev)
        rng[0] = 6.0
        ytt = torch.empty(1024, 8, 2, hw, 16, dtype=torch.float16, device=dev)
        measure("k_conv1x1_h2 terms 256 -> 128 @56 (1024)", lambda: _lib.check(lib.mirx_conv1x1_bn_relu_split2h_terms(
            vp(xb), cin * hw, cin, vp(sc), vp(sh), vp(w1), vp(o1), vp(bias), 1024, hw, vp(ytt), vp(rng), 1.0, 0.0, 16.0, 0.0,
            vp(yinv), 0, None), "c"))
        return
    # distance GEMM: 8192 queries over 1M x 1024
    g = torch.Generator(device=dev).manual_seed(0)
    gal = torch.nn.functional.normalize(torch.randn(1_000_000, 1024, generator=g, device=dev), dim=1)
    ix = FlatIndex(1024, "COSINE", 0)
    ix.add(gal)
    q = torch.nn.functional.normalize(torch.randn(8192, 1024, generator=g, device=dev), dim=1)
    measure("search 8192 q x 1M x 1024 (k_gemm16)", lambda: ix.search(q, 10), ms=30)
    del ix, gal
    # token-major Linear (DINOv2 qkv)
    m, k, n = 43840, 768, 2304
    x = torch.randn(m, k, device=dev)
    w3 = _split3_weights(torch.randn(n, k, device=dev) * k ** -0.5)
    y = torch.empty(m, n, device=dev)
    measure("k_linear_s3 43840 x 768 -> 2304", lambda: _lib.check(lib.mirx_linear_split3(vp(x), m, k, vp(w3), None, n, 0, None, None, vp(y), None), "l"))
    measure("torch F.linear fp32 (rocBLAS) same shape", lambda: torch.nn.functional.linear(x, torch.empty(n, k, device=dev).normal_()), ms=8)
    # attention
    qkv = torch.randn(32, 1370, 3, 12, 64, device=dev)
    o = torch.empty(32, 1370, 768, device=dev)
    measure("k_attention_s3 (32 x 1370 x 12 x 64)", lambda: _lib.check(lib.mirx_attention_qkv_f32_split3(vp(qkv), 32, 1370, 12, 64, 0.125, vp(o), None), "a"))
    measure("k_attention fp32 MFMA, same shape", lambda: _lib.check(lib.mirx_attention_qkv_f32(vp(qkv), 32, 1370, 12, 64, 0.125, vp(o), None), "a"))
    # Winograd fp32 56x56
    xx = torch.relu(torch.randn(1024, 128, 56, 56, device=dev))
    u = _winograd_weights(torch.randn(32, 128, 3, 3, device=dev) * 0.05)
    oo = torch.empty(1024, 32, 56, 56, device=dev)
    measure("k_conv3x3_wino<56> fp32 MFMA (1024 images)", lambda: _lib.check(lib.mirx_conv3x3_winograd_nchw(vp(xx), vp(u), 1024, 56, vp(oo), 32 * 3136, None), "c"))
    del xx, oo
    # whole DenseNet forward
    net = DenseNet121().eval().to(dev)
    img = torch.randn(1024, 3, 224, 224, device=dev)
    with torch.no_grad():
        measure("DenseNet-121 forward, 1024 images", lambda: net(img), ms=60)


if __name__ == "__main__":
    main()
