#!/usr/bin/env python3
"""Per-launch timeline of ONE DenseNet forward from a rocprofv3 --kernel-trace CSV (development tool).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python tools/bench_embed.py --batch 4096 --iters 2 --warmup 1
    python tools/layer_times.py gpurun_out/kt [--forward -1]

Splits the trace at k_range_absmax (the first launch of a two-fp16-term forward), prints every launch of the chosen
forward (duration, gap to the previous launch's end) and totals per kernel family and per dense block."""
import argparse
import collections
import csv
import glob
import re


def short(name):
    name = name.replace("void ", "").replace("mirx::(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", name)[:44]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--forward", type=int, default=-1)
    ap.add_argument("--all", action="store_true", help="print every launch")
    a = ap.parse_args()
    f = sorted(glob.glob(f"{a.dir}/**/*kernel_trace.csv", recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "k_range_absmax" in r["Kernel_Name"]]
    s = starts[a.forward]
    e = starts[a.forward + 1] if a.forward + 1 < len(starts) and a.forward != -1 else len(rows)
    fw = rows[s:e]
    # cut at the head kernel
    for i, r in enumerate(fw):
        if "k_head" in r["Kernel_Name"]:
            fw = fw[:i + 1]
            break
    t0 = int(fw[0]["Start_Timestamp"])
    fam = collections.defaultdict(lambda: [0.0, 0])
    prev_end, gaps = None, 0.0
    blk, blocks = 0, collections.defaultdict(lambda: collections.defaultdict(float))
    for r in fw:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        d = (en - st) / 1e3
        gap = 0.0 if prev_end is None else (st - prev_end) / 1e3
        gaps += max(gap, 0.0)
        prev_end = en
        nm = short(r["Kernel_Name"])
        m = re.search(r"k_conv3x3_d2p<(\d+)", nm)
        if m:
            blk = {56: 1, 28: 2, 14: 3, 7: 4}[int(m.group(1))]
        key = "conv1x1" if "k_conv1x1_h2<true, true, true" in nm else "conv3x3" if m else nm
        blocks[blk if key in ("conv1x1", "conv3x3") else 0][key] += d
        fam[nm][0] += d
        fam[nm][1] += 1
        if a.all:
            print(f"{(st - t0) / 1e3:10.1f} us  +{gap:7.1f}  {d:9.1f} us  {nm}  grid={r.get('Grid_Size_X', '')}")
    total = (int(fw[-1]["End_Timestamp"]) - t0) / 1e3
    busy = sum(v[0] for v in fam.values())
    print(f"forward: {len(fw)} launches, wall {total:.1f} us, kernels {busy:.1f} us, gaps {gaps:.1f} us")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        print(f"  {v[0]:10.1f} us {v[1]:4d} x  avg {v[0] / v[1]:8.1f}  {k}")
    # NOTE conv1x1 launches are attributed to the block of the conv3x3 that FOLLOWS them (first layer of a block: previous)
    for b in sorted(blocks):
        print(f"  block {b}: " + ", ".join(f"{k} {v:.1f}" for k, v in blocks[b].items()))


if __name__ == "__main__":
    main()
