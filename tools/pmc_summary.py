#!/usr/bin/env python3
"""Join rocprofv3 --pmc counter CSVs (one pass per counter group) with a --kernel-trace CSV and print
per-kernel HBM-side traffic and GB/s.  FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE
under-counts wide coalesced reads by 2x (MI355X_MICROARCH.md), so both the raw and the doubled read
figure are printed.  Usage: pmc_summary.py <fetch_counter.csv> <write_counter.csv> <kernel_trace.csv>"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("mirx::(anonymous namespace)::", "mirx::")
    name = re.sub(r"\(.*", "", name)
    return name[:70]


def counters(path, want):
    acc = defaultdict(float)
    n = defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == want:
                acc[short(r["Kernel_Name"])] += float(r["Counter_Value"])
                n[short(r["Kernel_Name"])] += 1
    return acc, n


def main():
    fetch, nf = counters(sys.argv[1], "FETCH_SIZE")
    write, _ = counters(sys.argv[2], "WRITE_SIZE")
    dur = defaultdict(float)
    with open(sys.argv[3]) as f:
        for r in csv.DictReader(f):
            dur[short(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    print("kernel,launches,time_ms,fetch_GB_raw,fetch_GB_x2,write_GB,GBps_raw,GBps_x2")
    for k in sorted(dur, key=lambda k: -dur[k]):
        if k not in fetch:
            continue
        fr, w, t = fetch[k] * 1024 / 1e9, write.get(k, 0.0) * 1024 / 1e9, dur[k]
        print(f"\"{k}\",{nf[k]},{t*1e3:.3f},{fr:.3f},{2*fr:.3f},{w:.3f},{(fr+w)/t:.0f},{(2*fr+w)/t:.0f}")


if __name__ == "__main__":
    main()
