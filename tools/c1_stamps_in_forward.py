"""Cycle stamps and the clock of k_conv1x1_h2 INSIDE a full DenseNet forward (development probe; needs the diagnostic library
built by `make -C image-retrieval---thesis-2026_amd/csrc diag`: MIRX_LIB_PATH=exp/libc1_st.so).  tools/bench_conv1x1.py --stamps
measures the same kernel with the layers of one block back to back; this shows what it holds when the whole forward runs."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirx import _lib                                   # noqa: E402
from mirx.model import DenseNet121                      # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = DenseNet121().eval().to(dev)
    x = torch.randn(4096, 3, 224, 224, device=dev)
    streams = [torch.cuda.Stream() for _ in range(2)]
    parts = list(x.chunk(2))

    def fwd():
        cur = torch.cuda.current_stream()
        for s_, p_ in zip(streams, parts):
            s_.wait_stream(cur)
            with torch.cuda.stream(s_):
                m(p_)
        for s_ in streams:
            cur.wait_stream(s_)

    with torch.no_grad():
        for _ in range(3):
            fwd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            fwd()
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 6
    dbg = ctypes.CDLL(_lib.LIB_PATH)
    sb = np.zeros(8192 * 8, dtype=np.uint64)
    assert dbg.mirx_debug_c1_stamps(sb.ctypes.data_as(ctypes.c_void_p)) == 0
    t8 = sb.reshape(8192, 8)
    valu = (t8[:, 7] & np.uint64(0xffffffff)).astype(np.float64)
    t = t8.astype(np.float64)
    t[:, 7] = (t8[:, 7] >> np.uint64(32)).astype(np.float64)
    keep = t[:, 2] > 0
    t, valu = t[keep], valu[keep]
    nk = t[:, 2]
    md = lambda v: float(np.median(v))      # noqa: E731
    print(f"forward of 4096 images on two streams: {dt * 1e3:.1f} ms = {4096 / dt:.0f} img/s (stamped build)")
    print(f"k_conv1x1_h2 stamps (whatever launches wrote last, {len(t)} workgroups, stages per workgroup median {md(nk):.0f}): "
          f"{md(t[:, 0] / nk):.0f} cycles/stage = wait {md(t[:, 3] / nk):.0f} + barrier {md(t[:, 4] / nk):.0f} + issue {md(t[:, 5] / nk):.0f} "
          f"+ frags/mfma {md(t[:, 6] / nk):.0f} + BN / ReLU / split {md(valu / nk):.0f} + LDS stores and the rest {md(t[:, 7] / nk):.0f}; clock median {md(t[:, 0] / t[:, 1] * 100):.0f} MHz, "
          f"10th / 90th percentile {float(np.percentile(t[:, 0] / t[:, 1] * 100, 10)):.0f} / {float(np.percentile(t[:, 0] / t[:, 1] * 100, 90)):.0f}")


if __name__ == "__main__":
    main()
