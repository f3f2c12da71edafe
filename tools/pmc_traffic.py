#!/usr/bin/env python3
"""Summarise the PMC passes of tools/pmc_traffic.sh: bytes = FETCH_SIZE(KB) * 1024 * 2 + WRITE_SIZE(KB) * 1024 per kernel
family (the gfx950 wide-load correction of MI355X_MICROARCH.md), per launch / per image / per forward."""
import csv
import glob
import json
import re
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("mirx::(anonymous namespace)::", "mirx::")
    return re.sub(r"\(.*", "", name)


def load(tag, ctr):
    acc, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(f"gpurun_out/pmct_{tag}_{ctr}/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == ctr:
                    acc[short(r["Kernel_Name"])] += float(r["Counter_Value"])
                    n[short(r["Kernel_Name"])] += 1
    return acc, n


def family(tag, pattern):
    f, nf = load(tag, "FETCH_SIZE")
    w, _ = load(tag, "WRITE_SIZE")
    ks = [k for k in f if re.search(pattern, k)]
    fetch = sum(f[k] for k in ks) * 1024.0
    write = sum(w.get(k, 0.0) for k in ks) * 1024.0
    return 2.0 * fetch + write, sum(nf[k] for k in ks), fetch, write


def dominant(tag):
    f, nf = load(tag, "FETCH_SIZE")
    w, _ = load(tag, "WRITE_SIZE")
    tot = {k: 2048.0 * f[k] + 1024.0 * w.get(k, 0.0) for k in f}
    k = max(tot, key=tot.get)
    return k, sum(tot.values())


def main():
    out = {"_batches": "passes run at the batch sizes bench.py times (DenseNet 2048 per stream, extras 64 / 32 / 16)",
           "_method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); bytes = 2 * FETCH_SIZE_KB * 1024 + "
                      "WRITE_SIZE_KB * 1024 (gfx950: FETCH_SIZE counts half the bytes of wide coalesced loads)"}
    b, n, fe, wr = family("gemm", r"k_gemm16<0, false>")
    if n:
        out["gemm_1Mx1024_q4096"] = {"bytes_per_launch": b / n, "launches": n, "fetch_bytes_raw": fe / n, "write_bytes": wr / n,
                                      "algorithmic_bytes": 1_000_000 * 1024 * 2 + 4096 * 1024 * 2}
    forwards, batch = 2, 2048                      # bench_embed --batch 2048 --iters 1 --warmup 1: the bench's forward per stream
    b, n, fe, wr = family("densenet", r"k_conv1x1_h2")
    if n:
        out["densenet121_conv1x1"] = {"bytes_per_image": b / (forwards * batch), "launches_per_forward": n // forwards,
                                       "fetch_bytes_raw_per_image": fe / (forwards * batch), "write_bytes_per_image": wr / (forwards * batch)}
    b3, n3, _, _ = family("densenet", r"k_conv3x3")
    if n3:
        out["densenet121_conv3x3"] = {"bytes_per_image": b3 / (forwards * batch), "launches_per_forward": n3 // forwards}
    k, tot = dominant("densenet") if n else (None, 0.0)
    if n:
        out["densenet121_forward"] = {"bytes_per_image": tot / (forwards * batch), "dominant_kernel": k,
                                      "algorithmic_bytes_per_image_fp32": 95.2e6}
    for tag, name, bs in (("convnextv2", "convnextv2_base_384", 64), ("dinov2", "dinov2_vitb14_518", 32), ("medsiglip", "medsiglip_448", 16)):
        try:
            k, tot = dominant(tag)
        except ValueError:
            continue
        out[name] = {"bytes_per_forward": tot / 2, "bytes_per_image": tot / (2 * bs), "batch": bs, "dominant_kernel": k}   # the batch bench.py times
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
