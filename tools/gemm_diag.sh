#!/bin/bash
# usage: tools/gemm_diag.sh <d> <lib> [<lib> ...]  -- device time of the filter GEMM per library build (rocprofv3 kernel trace;
# for the "results wrong" diagnostic builds, whose searches take follow-up passes that the stage timer would add in)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
d=$1; shift
for lib in "$@"; do
  tag=$(basename $lib .so)
  rm -rf gpurun_out/gd_$tag
  MIRX_LIB_PATH=$GRAFT_REPO_ROOT/$lib rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gd_$tag -- python tools/bench_search.py --d $d --q 4096 --iters 6 > gpurun_out/gd_$tag.log 2>&1
  python - <<PY
import csv, glob
rows = []
for f in glob.glob("gpurun_out/gd_$tag/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_gemm16<0" in r["Kernel_Name"]:
            rows.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
big = [x for x in rows if x > 0.5 * max(rows)] if rows else []
print(f"d=$d $tag: filter GEMM launches {len(rows)}, full-size {len(big)}: median {sorted(big)[len(big)//2] if big else -1:.3f} ms, min {min(big) if big else -1:.3f}")
PY
  rm -rf gpurun_out/gd_$tag
done
