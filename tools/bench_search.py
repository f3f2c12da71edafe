#!/usr/bin/env python3
"""Search-only micro benchmark (development tool): q/s and achieved MFMA TFLOP/s of the
tier-1 pass for a resident random gallery.  Not the contract bench (see bench.py)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirx.index import FlatIndex  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=1024)
    ap.add_argument("--q", type=int, default=8192)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", default="COSINE")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--sweep", default="", help="comma list of query counts (SURVEY 8d: 1,64,1024,8192)")
    ap.add_argument("--json", default="", help="write the sweep as JSON here")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1234)
    ix = FlatIndex(a.d, a.metric, 0)
    ix.reserve(a.n)
    chunk = 1 << 17
    for s in range(0, a.n, chunk):
        m = min(chunk, a.n - s)
        ix.add(torch.nn.functional.normalize(torch.randn(m, a.d, generator=g, device=dev), dim=1))
    if a.sweep:
        sweep(ix, a, dev)
        return
    q = torch.nn.functional.normalize(
        torch.randn(a.q, a.d, generator=torch.Generator(device=dev).manual_seed(4321), device=dev), dim=1)
    from mirx import _lib
    ix.set_option(_lib.OPT_PROFILE, 1)
    ix.search(q, a.k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        s, i = ix.search(q, a.k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    st = ix.last_stats()
    tm = ix.last_timings()
    flops = 2.0 * a.q * a.n * a.d
    print(f"N={a.n} D={a.d} Q={a.q} k={a.k} {a.metric}: {dt*1e3:.2f} ms/search, {a.q/dt:.0f} q/s, "
          f"{flops/dt/1e12:.1f} TFLOP/s end-to-end of the search call; gemm {tm['gemm']:.2f} ms = "
          f"{flops/tm['gemm']/1e9:.0f} TFLOP/s; stages={ {k: round(v, 3) for k, v in tm.items()} } stats={st}")


def sweep(ix, a, dev):
    """Search-only latency/throughput at several query-batch sizes over the same resident gallery."""
    import json
    rows = []
    for nq in [int(v) for v in a.sweep.split(",")]:
        q = torch.nn.functional.normalize(
            torch.randn(nq, a.d, generator=torch.Generator(device=dev).manual_seed(4321), device=dev), dim=1)
        for _ in range(3):
            ix.search(q, a.k)
        torch.cuda.synchronize()
        iters = max(a.iters, 3)
        t0 = time.perf_counter()
        for _ in range(iters):
            ix.search(q, a.k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        st = ix.last_stats()
        rows.append({"queries": nq, "ms_per_call": dt * 1e3, "queries_per_s": nq / dt,
                     "gallery_GBps_bf16": a.n * a.d * 2 / dt / 1e9, "tflops": 2.0 * nq * a.n * a.d / dt / 1e12,
                     "tier1_answered": st["tier1_answered"], "exact_answered": st["exact_answered"]})
        print(rows[-1], flush=True)
    if a.json:
        with open(a.json, "w") as f:
            json.dump({"gallery_rows": a.n, "dim": a.d, "k": a.k, "metric": a.metric, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
