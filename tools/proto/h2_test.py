"""Prototype measurement: fp16 two-term Linear (tools/proto/k_linear_h2.hip) vs the shipped bf16 three-term kernel:
speed, error against float64, sustained clock (needs gpurun_out/libh2.so and gpurun_out/libprobe.so)."""
import sys, ctypes, time, math, torch
sys.path.insert(0, '.')
from mirx import _lib
from mirx.model import _split3_weights
lib = _lib.load(); dev = torch.device("cuda:0")
h2 = ctypes.CDLL("gpurun_out/libh2.so")
h2.exp_linear_h2.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
vp = lambda t: ctypes.c_void_p(t.data_ptr())
def split2(w):
    wh = w.half(); wl = (w - wh.float()).half()
    n, k = w.shape
    t = torch.stack([wh, wl], 0).reshape(2, n // 128, 128, k // 16, 16)
    return t.permute(1, 3, 0, 2, 4).contiguous()
probe = None
try:
    probe = ctypes.CDLL("gpurun_out/libprobe.so"); probe.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p]
except OSError:
    pass
side = torch.cuda.Stream(); pout = torch.zeros(2, dtype=torch.int64, device=dev)
for (m, k, n) in ((43840, 768, 2304), (43840, 3072, 768), (4096, 768, 768)):
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(m, k, generator=g) * 1.5).to(dev); w = (torch.randn(n, k, generator=g) / math.sqrt(k)).to(dev)
    b = torch.randn(n, generator=g).to(dev)
    want = x[:2048].double() @ w.double().t() + b.double()
    y3 = torch.empty(m, n, device=dev); w3 = _split3_weights(w)
    _lib.check(lib.mirx_linear_split3(vp(x), m, k, vp(w3), vp(b), n, 0, None, None, vp(y3), None), "l")
    e3 = float((y3[:2048].double() - want).abs().max())
    for scale_w in (1.0, 1024.0):
        w2 = split2(w * scale_w); bs = b * scale_w
        y = torch.empty(m, n, device=dev)
        rc = h2.exp_linear_h2(vp(x), m, k, vp(w2), vp(bs), n, vp(y), None); assert rc == 0
        torch.cuda.synchronize()
        e = float(((y[:2048].double() / scale_w) - want).abs().max())
        run = lambda: h2.exp_linear_h2(vp(x), m, k, vp(w2), vp(bs), n, vp(y), None)
        for _ in range(3): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(50):
            run()
            if i == 2 and probe is not None: probe.clock_probe_launch(vp(pout), 3000, ctypes.c_void_p(side.cuda_stream))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        c, r = (int(v) for v in pout.cpu())
        print(f"m={m} k={k} n={n} w-scale {scale_w:6.0f}: fp16x2 {dt*1e3:.3f} ms {2.0*m*k*n/dt/1e12:6.1f} TF-eq  max err {e:.2e} (bf16x3: {e3:.2e}, |y|max {float(want.abs().max()):.1f})  clock {100.0*c/max(r,1):.0f} MHz", flush=True)
