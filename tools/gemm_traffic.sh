#!/bin/bash
# usage: tools/gemm_traffic.sh <lib> [<lib> ...]  -- fabric traffic (FETCH_SIZE x 2 + WRITE_SIZE, separate PMC passes) and device time of
# the filter GEMM per library build: the A/B behind the query-group size of the persistent schedule (MIRX_PLAN_GROUP)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  tag=$(basename $lib .so)
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/gt_${tag}_$ctr
    MIRX_LIB_PATH=$GRAFT_REPO_ROOT/$lib rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/gt_${tag}_$ctr -- python tools/bench_search.py --q 4096 --iters 1 > gpurun_out/gt_${tag}_$ctr.log 2>&1
  done
  python - <<PY
import csv, glob
def tot(ctr):
    v = []
    for f in glob.glob("gpurun_out/gt_${tag}_%s/**/*counter_collection.csv" % ctr, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_gemm16<0, false>" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                v.append(float(r["Counter_Value"]))
    return sum(v) / max(1, len(v)), len(v)
f, n = tot("FETCH_SIZE"); w, _ = tot("WRITE_SIZE")
print(f"$tag: {n} launches, fetch x2 {2 * f * 1024 / 1e9:.2f} GB + write {w * 1024 / 1e9:.3f} GB = {(2 * f + w) * 1024 / 1e9:.2f} GB per launch (algorithmic 2.06 GB)")
PY
  rm -rf gpurun_out/gt_${tag}_FETCH_SIZE gpurun_out/gt_${tag}_WRITE_SIZE
done
