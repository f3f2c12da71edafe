#!/bin/bash
# usage: ab.sh tag [lib]  -> bench-regime GEMM ms + sustained search GEMM ms
tag=$1; lib=$2
cd $GRAFT_REPO_ROOT
[ -n "$lib" ] && export MIRX_LIB_PATH=$GRAFT_REPO_ROOT/exp/$lib
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab_$tag.json 2>/dev/null
s=$(python tools/bench_search.py --iters 5 2>&1 | tail -1 | grep -o "gemm [0-9.]* ms")
python - <<PY
import json
d=json.load(open("gpurun_out/ab_$tag.json")); print("$tag", "bench gemm ms", round(d["roofline"]["avg_launch_ms"],3), "frac", round(d["roofline"]["frac"],4), "| sustained $s")
PY
