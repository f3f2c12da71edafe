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
