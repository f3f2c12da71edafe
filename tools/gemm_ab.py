#!/usr/bin/env python3
"""Same-box A/B of the filter GEMM across library builds (development tool).

    python tools/gemm_ab.py --libs image-retrieval---thesis-2026_amd/libmirx.so exp/libmirx_r03.so --dims 1024 256 512 --rounds 2

Every (library, width) runs tools/bench_search.py in a process of its own (MIRX_LIB_PATH), the libraries alternating inside
a round; prints the filter GEMM's HIP-event time per launch (median over rounds) and its fraction of the bf16 MFMA peak."""
import argparse
import os
import re
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 2516.6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--dims", nargs="+", type=int, default=[1024])
    ap.add_argument("--q", type=int, default=4096)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--metric", default="COSINE")
    a = ap.parse_args()
    res = {}
    for d in a.dims:
        for r in range(a.rounds):
            for lib in a.libs:
                env = dict(os.environ, MIRX_LIB_PATH=os.path.join(ROOT, lib))
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_search.py"), "--d", str(d), "--q", str(a.q),
                                      "--n", str(a.n), "--iters", str(a.iters), "--metric", a.metric], env=env, capture_output=True, text=True)
                m = re.search(r"gemm ([0-9.]+) ms", out.stdout)
                if not m:
                    print(lib, d, "FAILED", out.stdout[-400:], out.stderr[-800:], flush=True)
                    continue
                ms = float(m.group(1))
                res.setdefault((d, lib), []).append(ms)
                print(f"d={d} round {r} {lib}: gemm {ms:.3f} ms  | {out.stdout.strip()[-220:]}", flush=True)
    print("---- median ms per launch, fraction of the bf16 peak")
    for (d, lib), v in res.items():
        dimp = (d + 127) // 128 * 128
        ms = statistics.median(v)
        print(f"d={d:5d} {lib:50s} {ms:7.3f} ms  frac {2.0 * a.q * a.n * dimp / (ms * 1e-3) / 1e12 / PEAK:.4f}  ({', '.join(f'{x:.3f}' for x in v)})")


if __name__ == "__main__":
    main()
